#!/usr/bin/env python3
"""Per-cycle kernel timings of the bench workload (development probe, GPU only)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fusion-sim_amd"))
sys.path.insert(0, ROOT)
import fusionpic as fp  # noqa: E402
from bench import scene_grids, synthetic_inputs  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--side", type=int, default=10000)
ap.add_argument("--grid", type=int, default=1024)
ap.add_argument("--cycles", type=int, default=12)
ap.add_argument("--sort-interval", type=int, default=0)
ap.add_argument("--sync-each", type=int, default=1)
ap.add_argument("--rng", default="reference")
ap.add_argument("--fuse", type=int, default=1)
ap.add_argument("--precision", default="fp32")
args = ap.parse_args()

spec = dict(radius=1.0, height=1.0, nr=args.grid, nz=args.grid, dt=2e-9, nparticles=args.side,
            particle_mass=1.67e-27, particle_charge=1.602e-19)
n = args.side ** 2
pos, vel, entropy, rand = synthetic_inputs(n, spec, 0x5EEDF051)
sink, pdf = scene_grids(args.grid, args.grid)
sim = fp.makeCylindricalParticlePusher(spec, precision=args.precision, sort_interval=args.sort_interval, rng=args.rng,
                                       fuse_deposit="census" if args.fuse == 2 else bool(args.fuse))
sim.set(position=pos, velocity=vel, sink_mask=sink, source_pdf=pdf)
if args.rng == "reference":
    sim.setRandomState(entropy, rand)
sim.addBZ(0.01)
sim.precalc()
sim.profile(True)
# push on unsorted particles
sim.resetStats(); sim.step(); s = sim.stats()
print("push unsorted: %.3f ms" % s["ms_push"])
prev = sim.stats()
for c in range(args.cycles):
    sim.precalc(); sim.step(); sim.density()
    if args.sync_each:
        s = sim.stats()
        print("cycle %2d push %.3f cell_sums %.3f stamp %.3f precalc %.3f bin %.3f passes %d spilled %d (%.2f%%)" % (
            c, s["ms_push"] - prev["ms_push"], s["ms_deposit"] - prev["ms_deposit"], s["ms_stamp"] - prev["ms_stamp"],
            s["ms_precalc"] - prev["ms_precalc"], s["ms_sort"] - prev["ms_sort"], s["sort_passes"],
            s["deposit_spilled"], 100.0 * s["deposit_spilled"] / n))
        prev = s
s = sim.stats()
print("final", s)
