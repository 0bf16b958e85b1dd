#!/bin/bash
# Round 4 probe: the (r,z) push with 768- and 1024-thread workgroups (the kernel holds 150 VGPRs: three waves per SIMD; its
# 152 KB of LDS allow one workgroup per CU, so 512 threads are 8 of 12 possible waves).  Headline bench line per build.
cd $GRAFT_REPO_ROOT
for TH in ${THS:-512 768 1024}; do
  rm -f fusion-sim_amd/build/fpic_api.o
  make -C fusion-sim_amd EXTRA_HIPFLAGS="-DFPIC_PUSH_THREADS=$TH" all > gpurun_out/probe_build.log 2>&1 || { tail -5 gpurun_out/probe_build.log; exit 1; }
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extensions --no-strong-c4 $BENCH_ARGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$TH threads: value %.4g  ms_per_step %.3f  avg launch %.3f ms  frac %.3f' % (d['value'], d['ms_per_step'], r.get('avg_launch_ms', 0), r['frac']))" || exit 1
done
rm -f fusion-sim_amd/build/fpic_api.o
make -C fusion-sim_amd all > gpurun_out/probe_build.log 2>&1
