#!/bin/bash
# Round 4 probe: the full-EM push at 4 waves per SIMD (128 VGPRs, some spills, 1024-thread workgroups) against 3 (166 / 149 VGPRs)
cd $GRAFT_REPO_ROOT
em() { python bench.py --only-em --c3-particles $3 --c3-grid $4 --em-precision $2 --steps 4 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['em']; p=d['kernel_ms_per_substep']['push_gather_current']; print('$1  push %.3f ms = %.1f ps per particle' % (p, 1e9*p/$3))"; }
for V in "-DFES_EM_WAVES=3" "-DFES_EM_WAVES=4 -DFES_EM_THREADS_F32=1024 -DFES_EM_THREADS_F64=1024"; do
  rm -f fusion-sim_amd/build/fes_api.o
  make -C fusion-sim_amd EXTRA_HIPFLAGS="$V" all > gpurun_out/probe_build.log 2>&1 || { tail -5 gpurun_out/probe_build.log; exit 1; }
  for P in fp32 fp64; do
    em "em $P 256^3 5e8 [$V]" $P 500000000 256 && em "em $P 512^3 1e9 [$V]" $P 1000000000 512 || exit 1
  done
done
rm -f fusion-sim_amd/build/fes_api.o
make -C fusion-sim_amd all > gpurun_out/probe_build.log 2>&1
