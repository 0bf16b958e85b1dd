#!/bin/bash
# Round 4 probe: the electrostatic push against the number of particles per cell on ideal data (scripts/ablate_push3.hip: tile
# populations are whole multiples of a lane's four slots, nothing drifts): what a TILE costs besides its particles, and how much of
# that is the window (ABL 4: no staging, no flush).
cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I fusion-sim_amd/csrc -I include scripts/ablate_push3.hip -o /tmp/ablate_push3 || exit 1
for PPC in ${PPCS:-4 8 15 30}; do /tmp/ablate_push3 256 $PPC 0 65536 short || exit 1; done
/tmp/ablate_push3 512 8 0 65536 short
