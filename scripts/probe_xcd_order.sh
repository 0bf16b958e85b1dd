#!/bin/bash
# Round 4 probe — of commit c. "XCD-aware work order ... behind FPIC_XCD_ORDER"; the switch measured nothing gained and was
# removed again (profiles/r04_xcd_order_and_staging.txt), so part (1) of this script needs that commit's library.
# (1) FPIC_XCD_ORDER (each XCD a contiguous eighth of a tiled launch's work list) on the full-EM push at
# configs[4]'s lattice and on the electrostatic push; (2) FES_ABL_EM bits 8 (no window staging / flush) and 4 (no current
# deposit) at that lattice: what the 512^3 lattice costs the push beyond its particle streams.  Rebuilds fes_api.o ON THE
# GPU BOX for (2) and restores the real build.     scripts/probe_xcd_order.sh > gpurun_out/xcd_order.txt
cd $GRAFT_REPO_ROOT
em() { python bench.py --only-em --c3-particles ${N:-1000000000} --c3-grid ${G:-512} --em-precision fp64 --steps 4 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['em']; print('$1  push %.3f ms  lattice %.3f ms  sub-step %.3f ms' % (d['kernel_ms_per_substep']['push_gather_current'], d['kernel_ms_per_substep']['fdtd_b_e_b'], d['ms_per_substep']))"; }
c3() { python bench.py --only-c3 --c3-particles ${N:-500000000} --c3-grid ${G:-256} --steps 16 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['c3']; print('$1  ' + json.dumps({k: d[k] for k in ('ms_per_substep','kernel_ms_per_substep') if k in d}))"; }
FPIC_XCD_ORDER=0 em "em 512^3 1e9 fp64, list order " &&
FPIC_XCD_ORDER=1 em "em 512^3 1e9 fp64, XCD order  " &&
FPIC_XCD_ORDER=0 c3 "c3 256^3 5e8 fp32, list order " &&
FPIC_XCD_ORDER=1 c3 "c3 256^3 5e8 fp32, XCD order  " &&
N=1000000000 G=512 FPIC_XCD_ORDER=0 c3 "c3 512^3 1e9 fp32, list order " &&
N=1000000000 G=512 FPIC_XCD_ORDER=1 c3 "c3 512^3 1e9 fp32, XCD order  " &&
for A in 8 12; do
  rm -f fusion-sim_amd/build/fes_api.o
  make -C fusion-sim_amd EXTRA_HIPFLAGS="-DFES_ABL_EM=$A" all > gpurun_out/probe_build.log 2>&1 || { tail -5 gpurun_out/probe_build.log; exit 1; }
  FPIC_XCD_ORDER=0 em "em 512^3 1e9 fp64, FES_ABL_EM=$A  " || exit 1
done
rm -f fusion-sim_amd/build/fes_api.o
make -C fusion-sim_amd all > gpurun_out/probe_build.log 2>&1
