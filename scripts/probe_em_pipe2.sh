#!/bin/bash
# Round 5 probe: FES_EM_PIPE=2 (E) against the build's default (D) on ONE box, after the EM suite has passed on E.
cd $GRAFT_REPO_ROOT
rm -f fusion-sim_amd/build/fes_api.o; make -C fusion-sim_amd EXTRA_HIPFLAGS="-DFES_EM_PIPE=2" all > gpurun_out/probe_build.log 2>&1 || { tail -5 gpurun_out/probe_build.log; exit 1; }
timeout -k 10 400 python -m pytest tests/test_gpu_em.py -x -q -m gpu 2>&1 | tail -2
VARIANTS="E D E D" bash scripts/probe_em_pipe.sh
