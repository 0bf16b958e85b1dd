import sys, time, numpy as np
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'fusion-sim_amd')]
import fusionpic as fp
import bench
for side, grid in ((400, (400, 800)), (316, (128, 128)), (1000, (512, 512))):
    spec = dict(radius=1.0, height=2.0, nr=grid[0], nz=grid[1], dt=2e-9, nparticles=side, particle_mass=1.67e-27, particle_charge=1.602e-19)
    n = side * side
    pos, vel, entropy, rand = bench.synthetic_inputs(n, spec, 1)
    sink, pdf = bench.scene_grids(*grid)
    sim = fp.makeCylindricalParticlePusher(spec)
    sim.set(position=pos, velocity=vel, sink_mask=sink, source_pdf=pdf); sim.setRandomState(entropy, rand); sim.addBZ(0.01); sim.precalc()
    for _ in range(20): sim.step(); sim.density()
    sim.sync(); sim.resetStats(); sim.profile(True)
    t0 = time.perf_counter(); K = 300
    for _ in range(K): sim.step(); sim.density()
    sim.sync(); el = time.perf_counter() - t0
    st = sim.stats()
    print("n=%d grid=%s: %.1f us per frame (wall), kernels: push %.1f us stamp %.1f us sort %.2f us; %.2e updates/s" % (n, grid, 1e6*el/K, 1e3*st["ms_push"]/K, 1e3*st["ms_stamp"]/K, 1e3*st["ms_sort"]/K, 2*n*K/el))
    sim.destroy()
