#!/usr/bin/env python3
"""Time library variants (development): same scene, per-launch push time of fresh in-place launches."""
import glob, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fusion-sim_amd")); sys.path.insert(0, ROOT)
import fusionpic as fp
from bench import scene_grids, synthetic_inputs
spec = dict(radius=1.0, height=1.0, nr=1024, nz=1024, dt=2e-9, nparticles=10000, particle_mass=1.67e-27, particle_charge=1.602e-19)
n = 10000 ** 2
pos, vel, entropy, rand = synthetic_inputs(n, spec, 0x5EEDF051)
sink, pdf = scene_grids(1024, 1024)
libs = [("default", fp.LIB_PATH)] + [(os.path.basename(p)[13:-3], p) for p in sorted(glob.glob(os.path.join(ROOT, "build_probe/variants/*.so")))]
for name, path in libs:
    lib = fp.load_library(path)
    sim = fp.CylindricalParticlePusher(spec, library=lib)
    sim.set(position=pos, velocity=vel, sink_mask=sink, source_pdf=pdf); sim.setRandomState(entropy, rand)
    sim.addBZ(0.01); sim.precalc(); sim.sort(); sim.profile(True)
    times = []
    prev = 0.0
    for c in range(4):
        sim.precalc(); sim.step(); sim.density()
        s = sim.stats(); times.append(s["ms_push"] - prev); prev = s["ms_push"]
    print("%-12s push per launch: %s" % (name, " ".join("%.3f" % t for t in times)), flush=True)
    sim.destroy()
