#!/usr/bin/env python3
"""First precalc() of a freshly uploaded box at BASELINE configs[3] (512^3, 1e9 e- + 1e9 p+): VERDICT r03 item 8.
Round 3: the flat deposit with global atomics, 369 ms.  Round 4: the staged two-level binning first, then the tiled deposit."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fusion-sim_amd")]
import torch
import bench
import fusionpic as fp

total, grid, world = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000_000, 512, 8
spec, L, vth, mi, qi = bench.c4_scene(total, grid, world)
ne = total // 2
one = fp.makeCylindricalParticlePusher(dict(spec, count=ne))
one.addSpecies(mi, qi, ne)
for sp in range(2):
    for r in range(world):
        p, v = bench.c4_rank_particles(r, world, sp, ne // world, L, vth, 1.0 if sp == 0 else mi / spec["particle_mass"], 0)
        one.setRange(r * (ne // world), position=p, velocity=v, species=sp)
        del p, v
torch.cuda.empty_cache()
one.sync()
one.profile(True)
t0 = time.perf_counter()
one.precalc()
one.sync()
t1 = time.perf_counter()
st = one.stats()
print({"first_precalc_ms": 1e3 * (t1 - t0), "ms_sort": st["ms_sort"], "ms_deposit": st["ms_deposit"], "ms_solve": st["ms_solve"], "sort_passes": st["sort_passes"]})
t0 = time.perf_counter()
one.precalc()
one.sync()
print({"second_precalc_ms": 1e3 * (time.perf_counter() - t0)})
