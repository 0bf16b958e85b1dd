#!/bin/bash
# The full-EM push against the number of particles per cell (7.45 = 1e9 on configs[4]'s 512^3 lattice, 15 = configs[4], 30 = the
# bench line): scripts/em_density.sh > gpurun_out/em_density.txt
cd $GRAFT_REPO_ROOT
em() { python bench.py --only-em --c3-particles $3 --c3-grid $4 --em-precision $2 --steps 4 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['em']; p=d['kernel_ms_per_substep']['push_gather_current']; print('$1  push %.3f ms = %.1f ps per particle  lattice %.3f ms  sub-step %.3f ms' % (p, 1e9*p/$3, d['kernel_ms_per_substep']['fdtd_b_e_b'], d['ms_per_substep']))"; }
for P in fp32 fp64; do
  em "em $P 256^3 1.25e8 ( 7.45 per cell)" $P 125000000 256 &&
  em "em $P 256^3 2.5e8  (14.9  per cell)" $P 250000000 256 &&
  em "em $P 256^3 5e8    (29.8  per cell)" $P 500000000 256 &&
  em "em $P 512^3 1e9    ( 7.45 per cell)" $P 1000000000 512 || exit 1
done
