#!/bin/bash
# Round 4 probe: what the out-of-window particles (FES_ABL_EM bit 16: skipped) cost the full-EM push per tile.
cd $GRAFT_REPO_ROOT
em() { python bench.py --only-em --c3-particles $3 --c3-grid $4 --em-precision $2 --steps 4 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['em']; p=d['kernel_ms_per_substep']['push_gather_current']; print('$1  push %.3f ms = %.1f ps per particle' % (p, 1e9*p/$3))"; }
for A in 16 0; do
  rm -f fusion-sim_amd/build/fes_api.o
  make -C fusion-sim_amd EXTRA_HIPFLAGS="-DFES_ABL_EM=$A" all > gpurun_out/probe_build.log 2>&1 || { tail -5 gpurun_out/probe_build.log; exit 1; }
  for P in fp32 fp64; do
    em "em $P 256^3 1.25e8 FES_ABL_EM=$A" $P 125000000 256 && em "em $P 256^3 5e8 FES_ABL_EM=$A" $P 500000000 256 || exit 1
  done
done
