import os, sys
import numpy as np
sys.path.insert(0, "fusion-sim_amd"); sys.path.insert(0, "oracle"); sys.path.insert(0, "tests")
import fusionpic as fp
ME, QE, MP = 9.109e-31, -1.602e-19, 1.67e-27
world, shape = 4, (72, 40, 96)
rng = np.random.default_rng(world)
n = 30000
L = (0.016, 0.016, 0.001 * shape[2])
spec = dict(radius=L[0], length_y=L[1], height=L[2], nr=shape[0], ny=shape[1], nz=shape[2], dt=5e-12, nparticles=0, count=n, particle_mass=ME, particle_charge=QE,
            geometry="cart3d", solver="poisson_fft", macro_weight=1e15 * np.prod(L) / n)
pos = rng.random((n, 3)) * L
vel = rng.normal(0, 0.02, (n, 3))
one = fp.makeCylindricalParticlePusher(spec)
one.set(position=pos, velocity=vel)
nzl = shape[2] // world
own = np.floor(pos[:, 2] / L[2] * shape[2]).astype(int) // nzl
order = np.argsort(own, kind="stable")
p, v, counts = pos[order], vel[order], np.bincount(own, minlength=world)
one.set(position=p, velocity=v)
ranks = []
first = 0
for r in range(world):
    s = fp.makeCylindricalParticlePusher(dict(spec, count=n))
    s.domainInit(r, world, ghost_planes=2, migrate_every=2)
    s.domainSet(p[first:first + counts[r]], v[first:first + counts[r]], first_id=first)
    first += counts[r]
    ranks.append(s)
g = fp.BoxGroup(ranks)
one.precalc(); g.precalc()
for frame in range(8):
    one.step(); g.step()
    f1 = one.readField(fp.F3_RHO_FIXED).reshape(shape[2], -1)
    for r, s in enumerate(ranks):
        fr = s.readField(fp.F3_RHO_FIXED).reshape(shape[2], -1)
        bad = [k for k in range(r * nzl, (r + 1) * nzl) if not np.array_equal(fr[k], f1[k])]
        print("frame", frame, "rank", r, "bad own planes", bad, "sum diff", int(fr[r*nzl:(r+1)*nzl].sum() - f1[r*nzl:(r+1)*nzl].sum()))
