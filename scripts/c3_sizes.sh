#!/bin/bash
# The electrostatic push at the bench line (256^3 / 5e8) and at configs[3]'s density on its lattice (512^3 / 1e9, one species)
cd $GRAFT_REPO_ROOT
c3() { python bench.py --only-c3 --c3-particles $2 --c3-grid $3 --steps 16 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['c3']; k=d['kernel_ms_per_substep']; print('$1  push %.3f ms  solve %.3f ms  sub-step %.3f ms  frac %.3f' % (k['push_gather_deposit'], k['poisson_solve'], d['ms_per_substep'], d['roofline']['frac']))"; }
c3 "c3 256^3 5e8 fp32" 500000000 256 && c3 "c3 256^3 1.25e8 fp32" 125000000 256 && c3 "c3 512^3 1e9 fp32" 1000000000 512
