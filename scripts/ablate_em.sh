#!/bin/bash
# Timing probes of em_push_tiles_kernel (FES_ABL_EM bits: 1 no B gather, 2 no E gather, 4 no current deposit in the
# common case): rebuilds fes_api.o with each setting ON THE GPU BOX, runs bench.py --only-em, restores the real build.
# scripts/ablate_em.sh <precision> <particles> <grid>
P=${1:-fp64}; N=${2:-500000000}; G=${3:-256}
cd $GRAFT_REPO_ROOT
for A in 0 1 3 4 7; do
  rm -f fusion-sim_amd/build/fes_api.o
  make -C fusion-sim_amd EXTRA_HIPFLAGS="-DFES_ABL_EM=$A" all > gpurun_out/ablate_em_build.log 2>&1 || { tail -5 gpurun_out/ablate_em_build.log; exit 1; }
  python bench.py --only-em --c3-particles $N --c3-grid $G --em-precision $P --steps 4 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['em']; print('FES_ABL_EM=$A  $P  push %.3f ms  fdtd %.3f ms' % (d['kernel_ms_per_substep']['push_gather_current'], d['kernel_ms_per_substep']['fdtd_b_e_b']))"
done
rm -f fusion-sim_amd/build/fes_api.o
make -C fusion-sim_amd all > gpurun_out/ablate_em_build.log 2>&1
