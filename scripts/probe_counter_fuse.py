"""Counter-RNG mode at C2: the frame with the per-cell sums fused into the push (fuse_deposit=True) against the shipped
choice for this mode (tile census in the push, sums in their own pass)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fusion-sim_amd")]
import numpy as np
import fusionpic as fp
import bench
spec = dict(radius=1.0, height=1.0, nr=1024, nz=1024, dt=2e-9, nparticles=10000, particle_mass=1.67e-27, particle_charge=1.602e-19)
n = 10000 * 10000
pos, vel, entropy, rand = bench.synthetic_inputs(n, spec, 0x5EEDF051)
sink, pdf = bench.scene_grids(1024, 1024)
for fuse in ("census", True, False):
    sim = fp.makeCylindricalParticlePusher(spec, rng="counter", seed=0x5EEDF051, fuse_deposit=fuse)
    sim.set(position=pos, velocity=vel, sink_mask=sink, source_pdf=pdf)
    sim.addBZ(0.01); sim.precalc(); sim.sort()
    for _ in range(4):
        sim.precalc(); sim.step(); sim.density()
    sim.sync(); sim.resetStats(); sim.profile(True)
    t0 = time.perf_counter(); K = 24
    for _ in range(K):
        sim.precalc(); sim.step(); sim.density()
    sim.sync(); el = time.perf_counter() - t0
    st = sim.stats()
    print("fuse_deposit=%-8s %.3f ms per frame  push %.3f  cell sums %.3f  stamp %.3f  re-binnings %d  -> %.2e updates/s"
          % (fuse, 1e3 * el / K, st["ms_push"] / K, st["ms_deposit"] / K, st["ms_stamp"] / K, st["sort_passes"], 2 * n * K / el))
    sim.destroy()
