#!/bin/bash
# Round 4 probe: the (r,z) push's chunk (particles per work item) now that a workgroup has 768 threads x 4 particles = 3072 per turn:
# 16384 is 5.33 turns (the last one a third full), 12288 / 15360 / 18432 are 4 / 5 / 6 whole turns.
cd $GRAFT_REPO_ROOT
for C in ${CHUNKS:-16384 12288 15360 18432 24576}; do
  rm -f fusion-sim_amd/build/fpic_api.o fusion-sim_amd/build/fes_api.o fusion-sim_amd/build/fpic_host.o
  make -C fusion-sim_amd EXTRA_HIPFLAGS="-DFPIC_DEPOSIT_CHUNK=$C" all > gpurun_out/probe_build.log 2>&1 || { tail -5 gpurun_out/probe_build.log; exit 1; }
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extensions --no-strong-c4 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('chunk $C: value %.4g  ms_per_step %.3f  avg launch %.3f ms  frac %.3f' % (d['value'], d['ms_per_step'], r.get('avg_launch_ms', 0), r['frac']))" || exit 1
done
rm -f fusion-sim_amd/build/fpic_api.o fusion-sim_amd/build/fes_api.o fusion-sim_amd/build/fpic_host.o
make -C fusion-sim_amd all > gpurun_out/probe_build.log 2>&1
