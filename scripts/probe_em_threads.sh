#!/bin/bash
# Round 4 probe: the single-precision full-EM push with two 384-thread workgroups per CU instead of one of 768 (one stages
# or flushes its window while the other computes).  Rebuilds fes_api.o ON THE GPU BOX and restores the real build.
cd $GRAFT_REPO_ROOT
em() { python bench.py --only-em --c3-particles $2 --c3-grid $3 --em-precision fp32 --steps 4 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['em']; print('$1  push %.3f ms  lattice %.3f ms  sub-step %.3f ms' % (d['kernel_ms_per_substep']['push_gather_current'], d['kernel_ms_per_substep']['fdtd_b_e_b'], d['ms_per_substep']))"; }
for TH in ${THS:-768 384 512}; do
  rm -f fusion-sim_amd/build/fes_api.o
  make -C fusion-sim_amd EXTRA_HIPFLAGS="-DFES_EM_THREADS_F32=$TH" all > gpurun_out/probe_build.log 2>&1 || { tail -5 gpurun_out/probe_build.log; exit 1; }
  em "em fp32 256^3 5e8, $TH threads " 500000000 256 || exit 1
  em "em fp32 512^3 1e9, $TH threads " 1000000000 512 || exit 1
done
rm -f fusion-sim_amd/build/fes_api.o
make -C fusion-sim_amd all > gpurun_out/probe_build.log 2>&1
