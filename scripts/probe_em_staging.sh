#!/bin/bash
# Round 4 probe: FES_ABL_EM bits 8 (no window staging / flush) and 4 (no current deposit) on the full-EM push at 512^3 / 1e9,
# both precisions.  Rebuilds fes_api.o ON THE GPU BOX and restores the real build.
cd $GRAFT_REPO_ROOT
em() { python bench.py --only-em --c3-particles $3 --c3-grid $4 --em-precision $2 --steps 4 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['em']; print('$1  push %.3f ms  lattice %.3f ms  sub-step %.3f ms' % (d['kernel_ms_per_substep']['push_gather_current'], d['kernel_ms_per_substep']['fdtd_b_e_b'], d['ms_per_substep']))"; }
for A in 0 8 12; do
  rm -f fusion-sim_amd/build/fes_api.o
  make -C fusion-sim_amd EXTRA_HIPFLAGS="-DFES_ABL_EM=$A" all > gpurun_out/probe_build.log 2>&1 || { tail -5 gpurun_out/probe_build.log; exit 1; }
  for P in fp32 fp64; do
    em "em $P 512^3 1e9, FES_ABL_EM=$A " $P 1000000000 512 || exit 1
    em "em $P 256^3 1.25e8 (the same 7.45 per cell), FES_ABL_EM=$A " $P 125000000 256 || exit 1
  done
done
rm -f fusion-sim_amd/build/fes_api.o
make -C fusion-sim_amd all > gpurun_out/probe_build.log 2>&1
