#!/usr/bin/env python3
"""Development probe: per-step() kernel times of the CART3D cycle at BASELINE configs[2] size, by re-binning policy.
   python scripts/probe_c3.py [--particles 5e8] [--grid 256] [--intervals 0,2,4,6,8] [--steps 12]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "fusion-sim_amd"))
sys.argv_saved, sys.argv = sys.argv, sys.argv[:1]
import bench  # noqa: E402
sys.argv = sys.argv_saved

ap = argparse.ArgumentParser()
ap.add_argument("--particles", type=float, default=5e8)
ap.add_argument("--grid", type=int, default=256)
ap.add_argument("--intervals", default="0,2,4,6,8")
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--precision", default="fp32")
args = ap.parse_args()

import fusionpic as fp  # noqa: E402

n = int(args.particles)
for interval in [int(x) for x in args.intervals.split(",")]:
    spec, L, vth = bench.es3d_scene(n, args.grid)
    sim = fp.makeCylindricalParticlePusher(spec, sort_interval=interval, precision=args.precision)
    for first, pos, vel in bench.es3d_blocks(n, L, vth):
        sim.setRange(first, position=pos, velocity=vel)
    sim.sort(); sim.precalc(); sim.sync()
    sim.profile(True)
    rows, prev = [], sim.stats()
    for k in range(args.steps):
        sim.step()
        st = sim.stats()
        rows.append("%5.2f/%4.2f/%d" % ((st["ms_push"] - prev["ms_push"]) / 2, (st["ms_sort"] - prev["ms_sort"]) / 2, st["deposit_spilled"]))
        prev = st
    tot = (st["ms_push"] + st["ms_sort"] + st["ms_solve"]) / (2 * args.steps)
    print("sort_interval %d: %.3f ms per sub-step (push %.3f, solve %.3f, bin %.3f), %d re-binnings; per step() push/bin/spilled: %s"
          % (interval, tot, st["ms_push"] / (2 * args.steps), st["ms_solve"] / (2 * args.steps), st["ms_sort"] / (2 * args.steps),
             st["sort_passes"], " ".join(rows)), flush=True)
    sim.destroy()
