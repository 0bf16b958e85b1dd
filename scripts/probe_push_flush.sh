#!/bin/bash
# Round 4 probe: what the flush of a work item's sums window (up to 9 216 float atomics) costs the (r,z) push: FPIC_ABL_PUSH=1 skips it
# (timing only: the density is wrong).
cd $GRAFT_REPO_ROOT
for A in 0 1; do
  rm -f fusion-sim_amd/build/fpic_api.o
  make -C fusion-sim_amd EXTRA_HIPFLAGS="-DFPIC_ABL_PUSH=$A" all > gpurun_out/probe_build.log 2>&1 || { tail -5 gpurun_out/probe_build.log; exit 1; }
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extensions --no-strong-c4 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('FPIC_ABL_PUSH=$A: ms_per_step %.3f  avg launch %.3f ms' % (d['ms_per_step'], r.get('avg_launch_ms', 0)))" || exit 1
done
rm -f fusion-sim_amd/build/fpic_api.o
make -C fusion-sim_amd all > gpurun_out/probe_build.log 2>&1
