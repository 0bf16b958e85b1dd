"""GPU tests of the full-EM mode (spec.solver = 'yee'; BASELINE configs[4]) through the C ABI, against the build's own
oracle and the same known answers as tests/test_oracle_em.py.  PARITY UNPINNED (no reference counterpart).  Bar: for
given lattice fields, particles, the integer current grid and the updated E and B bit-exact against the oracle; the
lattice continuity equation exact in integers; Yee dispersion to 2e-5 (fp32) / 1e-11 (fp64) per step."""
import numpy as np
import pytest

from helpers import same_bits

pytestmark = pytest.mark.gpu

ME, QE = 9.109e-31, -1.602e-19
C = 2.998e8


@pytest.fixture(scope="module")
def fp():
    import fusionpic
    fusionpic.load_library()
    return fusionpic


@pytest.fixture(scope="module")
def eo():
    import es3d_oracle
    return es3d_oracle


def em_spec(shape, L, count, dt, **kw):
    s = dict(radius=L[0], length_y=L[1], height=L[2], nr=shape[0], ny=shape[1], nz=shape[2], dt=dt, nparticles=0, count=count,
             particle_mass=ME, particle_charge=QE, geometry="cart3d", solver="yee", macro_weight=1.0)
    s.update(kw)
    return s


def cfl_dt(shape, L, frac=0.5):
    d = [L[a] / shape[a] for a in range(3)]
    return frac / (C * np.sqrt(sum(1 / x ** 2 for x in d)))


def divergence(J, shape):
    J = J.reshape(shape[2], shape[1], shape[0], 3)
    return (J[..., 0] - np.roll(J[..., 0], 1, axis=2)) + (J[..., 1] - np.roll(J[..., 1], 1, axis=1)) + (J[..., 2] - np.roll(J[..., 2], 1, axis=0))


@pytest.mark.parametrize("precision", ["fp32", "fp64"])
@pytest.mark.parametrize("shape,n", [((8, 6, 10), 3001), ((20, 16, 12), 20000)])
def test_em_cycle_bit_exact_in_given_lattice_fields(fp, eo, precision, shape, n):
    """random E on the edges and B on the faces, two species, fast particles (every face-crossing pattern): after each
    of 6 sub-steps the particles, the integer current grid and both lattice fields are bit-identical to the oracle's,
    and 96 (rho^(n+1) - rho^n) + div J = 0 at every node"""
    dtype = np.float32 if precision == "fp32" else np.float64
    rng = np.random.default_rng(n)
    L = tuple(1e-3 * s for s in shape)
    spec = em_spec(shape, L, n, cfl_dt(shape, L), macro_weight=1e6)
    sim, ora = fp.makeCylindricalParticlePusher(spec, precision=precision), eo.OracleES3D(spec, dtype)
    ni = n // 5
    assert sim.addSpecies(1836 * ME, -2 * QE, ni) == ora.add_species(1836 * ME, -2 * QE, ni) == 1
    pos, vel = rng.random((n, 3)) * L, rng.normal(0, 0.3, (n, 3))
    pi, vi = rng.random((ni, 3)) * L, rng.normal(0, 0.05, (ni, 3))
    E, B = rng.normal(0, 1e4, shape + (3,)), rng.normal(0, 0.05, shape + (3,))
    for s in (sim, ora):
        s.set(position=pos, velocity=vel)
        s.set(position=pi, velocity=vi, species=1)
    sim.set(edge_E=E, face_B=B); ora.set_lattice(E=E, B=B)
    assert same_bits(sim.readField(fp.F3_E).ravel(), ora.E4) and same_bits(sim.readField(fp.F3_B_NODES).ravel(), ora.B4n)
    sim.density(); ora.deposit()
    assert np.array_equal(sim.readField(fp.F3_RHO_FIXED), ora.rho_fixed)
    for frame in range(3):
        before = ora.rho_fixed.copy()
        sim.step(); ora.step()
        for sp in (0, 1):
            got = sim.getParticles(species=sp)
            assert same_bits(got["position"], ora.positions(sp)) and same_bits(got["velocity"], ora.velocities(sp)), (frame, sp)
        assert np.array_equal(sim.readField(fp.F3_J_FIXED).ravel(), ora.J_fixed), frame
        assert same_bits(sim.readField(fp.F3_EDGE_E).ravel(), ora.Ey), frame
        assert same_bits(sim.readField(fp.F3_FACE_B).ravel(), ora.By), frame
        sim.density(); ora.deposit()
        fixed = sim.readField(fp.F3_RHO_FIXED)
        assert np.array_equal(fixed, ora.rho_fixed)
    sim.destroy()


@pytest.mark.parametrize("precision", ["fp32", "fp64"])
def test_continuity_exact_on_the_gpu(fp, eo, precision):
    """step() is two sub-steps and the current grid holds the last one's, so the identity is checked end to end: the GPU's
    charge grids before and after a step() differ by exactly minus the divergence of the two currents (the oracle's,
    to which the GPU's second one is compared bit for bit): 96 (rho_after - rho_before) + div (J1 + J2) = 0."""
    dtype = np.float32 if precision == "fp32" else np.float64
    rng = np.random.default_rng(5)
    shape, n = (10, 8, 6), 4000
    L = tuple(1e-3 * s for s in shape)
    spec = em_spec(shape, L, n, cfl_dt(shape, L), macro_weight=1e6)
    sim, ora = fp.makeCylindricalParticlePusher(spec, precision=precision), eo.OracleES3D(spec, dtype)
    pos, vel = rng.random((n, 3)) * L, rng.normal(0, 0.3, (n, 3))
    for s in (sim, ora):
        s.set(position=pos, velocity=vel)
    E, B = rng.normal(0, 1e4, shape + (3,)), rng.normal(0, 0.05, shape + (3,))
    sim.set(edge_E=E, face_B=B); ora.set_lattice(E=E, B=B)
    sim.density(); ora.deposit()
    r0 = sim.readField(fp.F3_RHO_FIXED).copy()
    assert np.array_equal(r0, ora.rho_fixed)
    div = np.zeros((shape[2], shape[1], shape[0]), dtype=np.int64)
    for _ in range(2):
        ora.em_substep()
        div += divergence(ora.J_fixed, shape)
    sim.step()
    sim.density(); ora.deposit()
    r1 = sim.readField(fp.F3_RHO_FIXED)
    assert np.array_equal(r1, ora.rho_fixed)
    assert not (96 * (r1 - r0).reshape(shape[2], shape[1], shape[0]) + div).any()
    assert np.array_equal(sim.readField(fp.F3_J_FIXED).ravel(), ora.J_fixed)   # the second sub-step's current
    sim.destroy()


@pytest.mark.parametrize("precision", ["fp32", "fp64"])
def test_vacuum_standing_wave_on_the_gpu(fp, eo, precision):
    shape, L, mode = (16, 12, 20), (0.16, 0.12, 0.2), (1, 2, 3)
    dt = cfl_dt(shape, L, 0.7)
    sim = fp.makeCylindricalParticlePusher(em_spec(shape, L, 1, dt, macro_weight=1e-30), precision=precision)
    sim.set(position=[[0.01, 0.01, 0.01]], velocity=[[0, 0, 0]])
    d = [L[a] / shape[a] for a in range(3)]
    idx = np.meshgrid(np.arange(shape[0]), np.arange(shape[1]), np.arange(shape[2]), indexing="ij")
    kk = [2 * np.pi * mode[a] / L[a] for a in range(3)]
    K = np.array([2 / d[a] * np.sin(kk[a] * d[a] / 2) for a in range(3)])
    e0 = np.cross(K, [0.3, -0.5, 0.8]); e0 = e0 / np.linalg.norm(e0) * 1e3
    E = np.zeros(shape + (3,))
    for a in range(3):
        E[..., a] = e0[a] * np.cos(sum(kk[b] * d[b] * (idx[b] + (0.5 if b == a else 0.0)) for b in range(3)))
    sim.set(edge_E=E, face_B=np.zeros(shape + (3,)))
    omega = 2 / dt * np.arcsin(C * dt * np.sqrt(sum((np.sin(kk[a] * d[a] / 2) / d[a]) ** 2 for a in range(3))))
    sim.step(20)
    got = sim.readField(fp.F3_EDGE_E, np.float64).reshape(shape[2], shape[1], shape[0], 4)[..., :3].transpose(2, 1, 0, 3)
    tol = 2e-5 if precision == "fp32" else 1e-11
    assert np.abs(got - E * np.cos(40 * omega * dt)).max() <= tol * 1e3 * 40
    sim.destroy()


@pytest.mark.parametrize("precision", ["fp32", "fp64"])
def test_em_from_precalc_keeps_gauss_and_energy(fp, eo, precision):
    """precalc(): the Poisson field on the edges + the uniform external B; 30 frames: div E - (rho - mean)/eps0 stays at
    rounding level, total energy within 1 %, and the run tracks the oracle's within the initial solve's tolerance"""
    dtype = np.float32 if precision == "fp32" else np.float64
    rng = np.random.default_rng(9)
    shape, L = (12, 12, 12), (0.012, 0.012, 0.012)
    n = 12 ** 3 * 8
    dt = cfl_dt(shape, L, 0.5)
    wp = 0.02 / dt
    dens = wp ** 2 * eo.EPS0 * ME / QE ** 2
    spec = em_spec(shape, L, n, dt, macro_weight=dens * np.prod(L) / n)
    sim, ora = fp.makeCylindricalParticlePusher(spec, precision=precision), eo.OracleES3D(spec, dtype)
    pos, vel = rng.random((n, 3)) * L, rng.normal(0, 0.05, (n, 3))
    for s in (sim, ora):
        s.set(position=pos, velocity=vel)
    sim.addB(0.0, 0.0, 0.01); ora.add_b(0.0, 0.0, 0.01)
    sim.precalc(); ora.precalc()

    def gauss(s):
        s.density()
        e = s.readField(fp.F3_EDGE_E, np.float64).reshape(shape[2], shape[1], shape[0], 4)
        dd = [L[a] / shape[a] for a in range(3)]
        div = ((e[..., 0] - np.roll(e[..., 0], 1, axis=2)) / dd[0] + (e[..., 1] - np.roll(e[..., 1], 1, axis=1)) / dd[1]
               + (e[..., 2] - np.roll(e[..., 2], 1, axis=0)) / dd[2])
        rho = s.readField(fp.F3_RHO_FIXED).reshape(shape[2], shape[1], shape[0]).astype(np.float64) * ora.rho_scale()
        return np.abs(div - (rho - rho.mean()) / eo.EPS0).max(), np.abs(rho).max() / eo.EPS0

    def energy(s):
        dv = np.prod(L) / np.prod(shape)
        e = s.readField(fp.F3_EDGE_E, np.float64)[:, :3]
        b = s.readField(fp.F3_FACE_B, np.float64)[:, :3]
        v = s.getParticles(np.float64)["velocity"]
        mu0 = 1 / (eo.EPS0 * C ** 2)
        return 0.5 * eo.EPS0 * (e ** 2).sum() * dv + 0.5 / mu0 * (b ** 2).sum() * dv + 0.5 * ME * spec["macro_weight"] * C ** 2 * (v ** 2).sum()

    tol = 3e-5 if precision == "fp32" else 1e-10
    res, scale = gauss(sim)
    assert res <= tol * scale
    e0 = energy(sim)
    sim.step(30); ora.step(30)
    res, scale = gauss(sim)
    assert res <= 60 * tol * scale
    assert abs(energy(sim) / e0 - 1) < 1e-2
    got = sim.getParticles(np.float64)
    dpos = np.abs(got["position"] - ora.positions()); dpos = np.minimum(dpos, 1 - dpos)
    assert dpos.max() <= (2e-3 if precision == "fp32" else 1e-7)
    sim.destroy()


@pytest.mark.parametrize("precision", ["fp32", "fp64"])
def test_em_cycle_with_the_staged_binning(fp, eo, monkeypatch, precision):
    """The LDS-staged two-level first binning (production: >= 2^20 particles) with the 8^3 tiles of the Yee mode."""
    monkeypatch.setenv("FPIC_TWO_LEVEL_MIN", "1")
    test_em_cycle_bit_exact_in_given_lattice_fields(fp, eo, precision, (40, 32, 48), 20000)


def test_em_cycle_on_a_grid_with_more_tiles_than_an_lds_histogram_holds(fp, eo):
    """288 x 288 x 280 nodes = 45 360 tiles of 8^3 cells (> 40 960: the census of the binning uses atomics on the
    global table and the scatter is the staged two-level one at any population size) — the path 512^3 grids take."""
    test_em_cycle_bit_exact_in_given_lattice_fields(fp, eo, "fp32", (288, 288, 280), 20000)


@pytest.mark.parametrize("precision", ["fp32", "fp64"])
@pytest.mark.parametrize("world,shape,ghost", [(2, (16, 12, 16), 2), (4, (24, 16, 32), 2), (4, (20, 20, 48), 4)])
def test_slab_decomposed_em_reproduces_one_gpu_bit_for_bit(fp, eo, precision, world, shape, ghost):
    """SURVEY 8(e) row 2 for the full-EM cycle (BASELINE configs[4] is its 8-GPU form): `world` ranks stepped as an
    in-process group — halo copies of the edge E and the face B (ghost + 2 planes per side), ghost-plane reduce of the
    int64 current, migration.  Against ONE handle holding everything, after every frame: the current grid, both lattice
    fields (on each rank's own planes) and every particle (matched by its global index) are bit-identical."""
    dtype = np.float32 if precision == "fp32" else np.float64
    rng = np.random.default_rng(world * 100 + ghost)
    n, ni = 24000, 6000
    L = tuple(1e-3 * s for s in shape)
    spec = em_spec(shape, L, n, cfl_dt(shape, L), macro_weight=1e6)
    nzl = shape[2] // world
    pos, vel = rng.random((n, 3)) * L, rng.normal(0, 0.3, (n, 3))          # fast: a cell every two or three sub-steps
    pi, vi = rng.random((ni, 3)) * L, rng.normal(0, 0.05, (ni, 3))
    sets = []
    for p, v in ((pos, vel), (pi, vi)):                                      # global indices contiguous per rank
        owner = np.floor(p[:, 2] / L[2] * shape[2]).astype(int) // nzl
        order = np.argsort(owner, kind="stable")
        sets.append((p[order], v[order], np.bincount(owner, minlength=world)))
    E, B = rng.normal(0, 1e4, shape + (3,)), rng.normal(0, 0.05, shape + (3,))

    one = fp.makeCylindricalParticlePusher(spec, precision=precision)
    one.addSpecies(1836 * ME, -2 * QE, ni)
    for sp, (p, v, _) in enumerate(sets):
        one.set(position=p, velocity=v, species=sp)
    one.set(edge_E=E, face_B=B)
    ranks = []
    for r in range(world):
        s = fp.makeCylindricalParticlePusher(spec, precision=precision)     # capacity: everything could end up here
        s.addSpecies(1836 * ME, -2 * QE, ni)
        s.domainInit(r, world, ghost_planes=ghost, migrate_every=2)
        for sp, (p, v, counts) in enumerate(sets):
            first = int(counts[:r].sum())
            s.domainSet(p[first:first + counts[r]], v[first:first + counts[r]], first_id=first, species=sp)
        s.set(edge_E=E, face_B=B)
        ranks.append(s)
    group = fp.BoxGroup(ranks)

    def compare(tag):
        for which in (fp.F3_EDGE_E, fp.F3_FACE_B):
            ref = one.readField(which).reshape(shape[2], -1)
            for r, s in enumerate(ranks):
                got = s.readField(which).reshape(shape[2], -1)
                assert same_bits(got[r * nzl:(r + 1) * nzl], ref[r * nzl:(r + 1) * nzl]), (tag, which, r)
        for sp, total in ((0, n), (1, ni)):
            parts = [s.domainGet(species=sp) for s in ranks]
            ids = np.concatenate([p["ids"] for p in parts])
            assert len(ids) == total and np.array_equal(np.sort(ids), np.arange(total)), (tag, sp)
            ref = one.getParticles(species=sp)
            assert same_bits(np.concatenate([p["position"] for p in parts])[np.argsort(ids)], ref["position"]), (tag, sp)
            assert same_bits(np.concatenate([p["velocity"] for p in parts])[np.argsort(ids)], ref["velocity"]), (tag, sp)

    compare("upload")
    for frame in range(4):
        one.step(); group.step()
        jref = one.readField(fp.F3_J_FIXED).reshape(shape[2], -1)
        for r, s in enumerate(ranks):
            got = s.readField(fp.F3_J_FIXED).reshape(shape[2], -1)
            assert np.array_equal(got[r * nzl:(r + 1) * nzl], jref[r * nzl:(r + 1) * nzl]), (frame, r)
        compare("frame %d" % frame)
        # density() of the frame loop (fusionsim.js:174) on a decomposed full-EM run: the charge grid of the current
        # positions, ghost planes exchanged and added — complete and exact on every rank's own planes
        one.density(); group.density()
        rref = one.readField(fp.F3_RHO_FIXED).reshape(shape[2], -1)
        for r, s in enumerate(ranks):
            got = s.readField(fp.F3_RHO_FIXED).reshape(shape[2], -1)
            assert np.array_equal(got[r * nzl:(r + 1) * nzl], rref[r * nzl:(r + 1) * nzl]), ("density", frame, r)
    stats = [s.domainStats() for s in ranks]
    assert sum(s["migrated"] for s in stats) > 0 and all(s["lost"] == 0 for s in stats)
    for s in ranks + [one]:
        s.destroy()


@pytest.mark.parametrize("precision", ["fp32", "fp64"])
def test_slab_decomposed_em_from_precalc(fp, eo, precision):
    """precalc() of a decomposed full-EM run: ghost-plane reduce of the charge, every rank solves the gathered grid, the
    Poisson field on the edges + the external B; then 6 frames: lattice fields on the own planes and all particles
    bit-identical to one handle's, Gauss's law still at rounding level on the assembled field."""
    rng = np.random.default_rng(17)
    world, shape, L = 2, (12, 12, 16), (0.012, 0.012, 0.016)
    n = 12 * 12 * 16 * 8
    dt = cfl_dt(shape, L, 0.5)
    dens = (0.02 / dt) ** 2 * eo.EPS0 * ME / QE ** 2
    spec = em_spec(shape, L, n, dt, macro_weight=dens * np.prod(L) / n)
    nzl = shape[2] // world
    pos, vel = rng.random((n, 3)) * L, rng.normal(0, 0.05, (n, 3))
    owner = np.floor(pos[:, 2] / L[2] * shape[2]).astype(int) // nzl
    order = np.argsort(owner, kind="stable")
    pos, vel, counts = pos[order], vel[order], np.bincount(owner, minlength=world)
    one = fp.makeCylindricalParticlePusher(spec, precision=precision)
    one.set(position=pos, velocity=vel)
    one.addB(0.0, 0.0, 0.01)
    ranks = []
    for r in range(world):
        s = fp.makeCylindricalParticlePusher(spec, precision=precision)
        s.domainInit(r, world, ghost_planes=2, migrate_every=8)
        first = int(counts[:r].sum())
        s.domainSet(pos[first:first + counts[r]], vel[first:first + counts[r]], first_id=first)
        s.addB(0.0, 0.0, 0.01)
        ranks.append(s)
    group = fp.BoxGroup(ranks)
    one.precalc(); group.precalc()

    def assembled(which):
        out = np.empty_like(one.readField(which)).reshape(shape[2], -1)
        for r, s in enumerate(ranks):
            out[r * nzl:(r + 1) * nzl] = s.readField(which).reshape(shape[2], -1)[r * nzl:(r + 1) * nzl]
        return out

    for frame in range(7):
        for which in (fp.F3_EDGE_E, fp.F3_FACE_B):
            assert same_bits(assembled(which), one.readField(which).reshape(shape[2], -1)), (frame, which)
        parts = [s.domainGet() for s in ranks]
        ids = np.concatenate([p["ids"] for p in parts])
        ref = one.getParticles()
        assert np.array_equal(np.sort(ids), np.arange(n))
        assert same_bits(np.concatenate([p["position"] for p in parts])[np.argsort(ids)], ref["position"]), frame
        assert same_bits(np.concatenate([p["velocity"] for p in parts])[np.argsort(ids)], ref["velocity"]), frame
        if frame < 6:
            one.step(); group.step()
    with pytest.raises(fp.FusionPicError):
        ranks[0].density()
    for s in ranks + [one]:
        s.destroy()


@pytest.mark.parametrize("precision", ["fp32", "fp64"])
@pytest.mark.parametrize("world,shape,ghost,start", [(4, (16, 16, 64), 2, "precalc"), (8, (16, 32, 128), 3, "precalc"), (4, (32, 16, 64), 1, "upload"),
                                                     (2, (16, 16, 32), 2, "precalc")])
def test_decomposed_em_with_slab_only_arrays(fp, eo, monkeypatch, precision, world, shape, ghost, start):
    """distributed_solve on the ranks of a full-EM run (power-of-two lattice: the library's own transforms): the initial
    field comes from the decomposed solve, whose phi is the one handle's bit for bit, and every rank keeps nzl + 2 H + 1
    planes (H = ghost + 2) of its node arrays instead of the whole lattice.  From precalc() or from uploaded fields, over
    frames with migration: the current grid, both lattice fields on the own planes and every particle bit-identical to one
    handle's; the same with whole-grid arrays (FPIC_DOMAIN_COMPACT=0); the grid memory of a rank shrinks."""
    rng = np.random.default_rng(23 + world)
    L = tuple(1e-3 * s for s in shape)
    n = shape[0] * shape[1] * shape[2] * 2
    dt = cfl_dt(shape, L, 0.5)
    dens = (0.02 / dt) ** 2 * eo.EPS0 * ME / QE ** 2
    spec = em_spec(shape, L, n, dt, macro_weight=dens * np.prod(L) / n)
    nzl = shape[2] // world
    pos, vel = rng.random((n, 3)) * L, rng.normal(0, 0.25, (n, 3))
    owner = np.floor(pos[:, 2] / L[2] * shape[2]).astype(int) // nzl
    order = np.argsort(owner, kind="stable")
    pos, vel, counts = pos[order], vel[order], np.bincount(owner, minlength=world)
    E, B = rng.normal(0, 1e4, shape + (3,)), rng.normal(0, 0.05, shape + (3,))

    def prepare(sim):
        if start == "upload":
            sim.set(edge_E=E, face_B=B)
        else:
            sim.addB(0.0, 0.0, 0.01)

    one = fp.makeCylindricalParticlePusher(spec, precision=precision)
    one.set(position=pos, velocity=vel)
    prepare(one)
    if start == "precalc":
        one.precalc()
    trace = []                                       # (edge E, face B, J, positions, velocities) of one handle, frame by frame
    def snapshot_one():
        p = one.getParticles()
        return [one.readField(w).reshape(shape[2], -1) for w in (fp.F3_EDGE_E, fp.F3_FACE_B, fp.F3_J_FIXED)] + [p["position"], p["velocity"]]
    trace.append(snapshot_one())
    frames = 5
    for _ in range(frames):
        one.step()
        trace.append(snapshot_one())
    one.destroy()

    grid_bytes = {}
    for layout in ("whole", "slab"):
        if layout == "whole":
            monkeypatch.setenv("FPIC_DOMAIN_COMPACT", "0")
        else:
            monkeypatch.delenv("FPIC_DOMAIN_COMPACT")
        ranks = []
        for r in range(world):
            s = fp.makeCylindricalParticlePusher(spec, precision=precision)
            s.domainInit(r, world, ghost_planes=ghost, migrate_every=2, distributed_solve=True)
            first = int(counts[:r].sum())
            s.domainSet(pos[first:first + counts[r]], vel[first:first + counts[r]], first_id=first)
            prepare(s)
            ranks.append(s)
        group = fp.BoxGroup(ranks)
        if start == "precalc":
            group.precalc()
        grid_bytes[layout] = [s.stats()["bytes_grid_state"] for s in ranks]
        for frame in range(frames + 1):
            ref = trace[frame]
            for r, s in enumerate(ranks):
                own = slice(r * nzl, (r + 1) * nzl)
                for w, which in enumerate((fp.F3_EDGE_E, fp.F3_FACE_B, fp.F3_J_FIXED)):
                    if which == fp.F3_J_FIXED and frame == 0:
                        continue
                    got = s.readField(which).reshape(shape[2], -1)[own]
                    assert (np.array_equal if which == fp.F3_J_FIXED else same_bits)(got, ref[w][own]), (layout, frame, r, which)
            parts = [s.domainGet() for s in ranks]
            ids = np.concatenate([p["ids"] for p in parts])
            assert np.array_equal(np.sort(ids), np.arange(n)), (layout, frame)
            assert same_bits(np.concatenate([p["position"] for p in parts])[np.argsort(ids)], ref[3]), (layout, frame)
            assert same_bits(np.concatenate([p["velocity"] for p in parts])[np.argsort(ids)], ref[4]), (layout, frame)
            if frame < frames:
                group.step()
        stats = [s.domainStats() for s in ranks]
        assert sum(st["migrated"] for st in stats) > 0 and all(st["lost"] == 0 for st in stats), layout
        group.density()                              # (the frame loop's density() on slab-only arrays)
        total = sum(int(s.readField(fp.F3_RHO_FIXED).reshape(shape[2], -1)[r * nzl:(r + 1) * nzl].astype(object).sum()) for r, s in enumerate(ranks))
        assert total == n * 2 ** 42, layout
        for s in ranks:
            s.destroy()
    nzs = nzl + 2 * (ghost + 2) + 1
    esz = 4 if precision == "fp32" else 8
    tile = 128 // (2 * esz)                          # the transform buffer's rows are padded to whole column tiles of 128 bytes
    saved = (shape[2] - nzs) * shape[0] * shape[1] * (8 + 24 + 18 * esz) + (shape[0] // 2 + tile) // tile * tile * shape[1] * shape[2] * 2 * esz
    assert nzs < shape[2] and all(w - c == saved for w, c in zip(grid_bytes["whole"], grid_bytes["slab"]))


def test_large_decomposed_em_is_bit_identical_to_one_handle(fp, eo):
    """1.2e7 electrons on a 64 x 64 x 128 Yee lattice over 8 in-process ranks (1.5e6 per rank: the staged two-level
    binning of the 8^3-cell tiles), migration every 4 sub-steps, from precalc(): after 4 frames the current grid and both
    lattice fields on every rank's planes and every particle are bit-identical to one handle's."""
    world, shape = 8, (64, 64, 128)
    L = (0.064, 0.064, 0.128)
    n = 12_000_000
    dt = cfl_dt(shape, L, 0.5)
    dens = (0.05 / dt) ** 2 * eo.EPS0 * ME / QE ** 2
    spec = em_spec(shape, L, n, dt, macro_weight=dens * np.prod(L) / n)
    rng = np.random.Generator(np.random.Philox(5))
    nzl = shape[2] // world
    z = np.sort(rng.random(n, dtype=np.float32)) * np.float32(L[2] * (1 - 1e-6))
    pos = np.stack([rng.random(n, dtype=np.float32) * np.float32(L[0]), rng.random(n, dtype=np.float32) * np.float32(L[1]), z], axis=1)
    vel = rng.standard_normal((n, 3), dtype=np.float32) * np.float32(0.2)          # 0.06 cells per sub-step
    counts = np.bincount(np.floor(z.astype(np.float64) / L[2] * shape[2]).astype(int) // nzl, minlength=world)
    one = fp.makeCylindricalParticlePusher(spec)
    one.set(position=pos, velocity=vel)
    one.addB(0.0, 0.0, 0.02)
    ranks = []
    for r in range(world):
        s = fp.makeCylindricalParticlePusher(dict(spec, count=int(counts.max() * 1.2)))
        s.domainInit(r, world, ghost_planes=2, migrate_every=4)
        first = int(counts[:r].sum())
        s.domainSet(pos[first:first + counts[r]], vel[first:first + counts[r]], first_id=first)
        s.addB(0.0, 0.0, 0.02)
        ranks.append(s)
    group = fp.BoxGroup(ranks)
    one.precalc(); group.precalc()
    for _ in range(4):
        one.step(); group.step()
    for which in (fp.F3_J_FIXED, fp.F3_EDGE_E, fp.F3_FACE_B):
        ref = one.readField(which).reshape(shape[2], -1)
        for r, s in enumerate(ranks):
            got = s.readField(which).reshape(shape[2], -1)[r * nzl:(r + 1) * nzl]
            assert (np.array_equal if which == fp.F3_J_FIXED else same_bits)(got, ref[r * nzl:(r + 1) * nzl]), (which, r)
    parts = [s.domainGet() for s in ranks]
    ids = np.concatenate([p["ids"] for p in parts])
    order = np.argsort(ids)
    assert np.array_equal(ids[order], np.arange(n, dtype=np.uint32))
    ref = one.getParticles()
    assert same_bits(np.concatenate([p["position"] for p in parts])[order], ref["position"])
    assert same_bits(np.concatenate([p["velocity"] for p in parts])[order], ref["velocity"])
    stats = [s.domainStats() for s in ranks]
    assert sum(s["migrated"] for s in stats) > 1000 and all(s["lost"] == 0 for s in stats)
    for s in ranks + [one]:
        s.destroy()


@pytest.mark.parametrize("seed", list(range(8)))
def test_em_decomposition_randomised(fp, eo, seed):
    """Random full-EM decompositions against one handle: world 2..6, 1..3 ghost planes, migration every 1..6 sub-steps,
    one or two species, uneven populations, either precision, random lattice fields; currents, fields and particles
    bit-identical after every frame."""
    rng = np.random.default_rng(2000 + seed)
    precision = "fp32" if rng.random() < 0.5 else "fp64"
    world = int(rng.choice([2, 3, 4, 6]))
    G = int(rng.integers(1, 4))
    every = int(rng.integers(1, 7))
    nzl = int(rng.integers(2 * (G + 2), 2 * (G + 2) + 6))
    shape = (int(rng.integers(6, 24)), int(rng.integers(6, 24)), world * nzl)
    L = tuple(1e-3 * s for s in shape)
    dt = cfl_dt(shape, L, 0.5)
    n = int(rng.integers(3000, 30000))
    spec = em_spec(shape, L, n, dt, macro_weight=1e6)
    vmax = min(0.9, 0.8 * G * 1e-3 / (every * dt * C))
    two = rng.random() < 0.5
    pops = []
    for m in ((n, n // 4) if two else (n,)):
        z = (rng.random(m) ** float(rng.choice([1.0, 3.0]))) * L[2] * (1 - 1e-9)
        p = np.stack([rng.random(m) * L[0], rng.random(m) * L[1], z], axis=1)
        v = np.stack([rng.normal(0, 0.2, m), rng.normal(0, 0.2, m), rng.uniform(-vmax, vmax, m)], axis=1)
        owner = np.floor(p[:, 2] / L[2] * shape[2]).astype(int) // nzl
        order = np.argsort(owner, kind="stable")
        pops.append((p[order], v[order], np.bincount(owner, minlength=world)))
    E, B = rng.normal(0, 1e4, shape + (3,)), rng.normal(0, 0.05, shape + (3,))

    def build(count, r=None):
        s = fp.makeCylindricalParticlePusher(dict(spec, count=count), precision=precision)
        if two:
            s.addSpecies(1836 * ME, -2 * QE, n // 4)
        if r is not None:
            s.domainInit(r, world, ghost_planes=G, migrate_every=every)
        return s

    one = build(n)
    for sp, (p, v, _) in enumerate(pops):
        one.set(position=p, velocity=v, species=sp)
    one.set(edge_E=E, face_B=B)
    ranks = []
    for r in range(world):
        s = build(4 * n, r)
        for sp, (p, v, c) in enumerate(pops):
            first = int(c[:r].sum())
            if c[r]:
                s.domainSet(p[first:first + c[r]], v[first:first + c[r]], first_id=first, species=sp)
        s.set(edge_E=E, face_B=B)
        ranks.append(s)
    group = fp.BoxGroup(ranks)
    for frame in range(4):
        one.step(); group.step()
        for which in (fp.F3_J_FIXED, fp.F3_EDGE_E, fp.F3_FACE_B):
            ref = one.readField(which).reshape(shape[2], -1)
            for r, s in enumerate(ranks):
                got = s.readField(which).reshape(shape[2], -1)[r * nzl:(r + 1) * nzl]
                assert (np.array_equal if which == fp.F3_J_FIXED else same_bits)(got, ref[r * nzl:(r + 1) * nzl]), (frame, which, r)
        for sp, (p, _, _) in enumerate(pops):
            parts = [s.domainGet(species=sp) for s in ranks]
            ids = np.concatenate([q["ids"] for q in parts])
            order = np.argsort(ids)
            assert np.array_equal(ids[order], np.arange(len(p))), (frame, sp)
            ref = one.getParticles(species=sp)
            assert same_bits(np.concatenate([q["position"] for q in parts])[order], ref["position"]), (frame, sp)
            assert same_bits(np.concatenate([q["velocity"] for q in parts])[order], ref["velocity"]), (frame, sp)
    assert all(s.domainStats()["lost"] == 0 for s in ranks)
    for s in ranks + [one]:
        s.destroy()


@pytest.mark.parametrize("seed", list(range(10)))
def test_em_randomised_against_the_oracle(fp, eo, monkeypatch, seed):
    """Random full-EM runs in random lattice fields against the oracle: grids from 3 x 3 x 3 to sizes that are no multiple
    of the 8^3 tile, 1..3 species, time steps up to the CFL limit, fast particles (every face-crossing pattern), either
    precision, the staged binning forced on every other case: particles, the integer current, both lattice fields
    bit-identical after every frame, and the lattice continuity equation exact."""
    rng = np.random.default_rng(5000 + seed)
    if seed % 2:
        monkeypatch.setenv("FPIC_TWO_LEVEL_MIN", "1")
    precision = "fp32" if rng.random() < 0.5 else "fp64"
    dtype = np.float32 if precision == "fp32" else np.float64
    shape = tuple(int(x) for x in rng.integers(3, [30, 26, 22]))
    L = tuple(float(x) for x in rng.uniform(0.5e-3, 2e-3, 3) * shape)
    n = int(rng.integers(1, 15000))
    spec = em_spec(shape, L, n, cfl_dt(shape, L, float(rng.uniform(0.2, 0.95))), macro_weight=float(rng.uniform(1e3, 1e7)))
    sim, ora = fp.makeCylindricalParticlePusher(spec, precision=precision), eo.OracleES3D(spec, dtype)
    counts = [n]
    for _ in range(int(rng.integers(0, 3))):
        m, z, mass = int(rng.integers(1, 4000)), int(rng.choice([-2, -1, 1, 2])), float(rng.choice([ME, 1836 * ME]))
        assert sim.addSpecies(mass, z * QE, m) == ora.add_species(mass, z * QE, m)
        counts.append(m)
    for sp, m in enumerate(counts):
        p, v = rng.random((m, 3)) * L, rng.normal(0, float(rng.uniform(0.01, 0.5)), (m, 3))
        sim.set(position=p, velocity=v, species=sp); ora.set(position=p, velocity=v, species=sp)
    E, B = rng.normal(0, 1e4, shape + (3,)), rng.normal(0, 0.05, shape + (3,))
    sim.set(edge_E=E, face_B=B); ora.set_lattice(E=E, B=B)
    sim.density(); ora.deposit()
    for frame in range(3):
        before = ora.rho_fixed.copy()
        sim.step(); ora.step()
        for sp in range(len(counts)):
            got = sim.getParticles(species=sp)
            assert same_bits(got["position"], ora.positions(sp)) and same_bits(got["velocity"], ora.velocities(sp)), (seed, frame, sp)
        J = sim.readField(fp.F3_J_FIXED).ravel()
        assert np.array_equal(J, ora.J_fixed), (seed, frame)
        assert same_bits(sim.readField(fp.F3_EDGE_E).ravel(), ora.Ey) and same_bits(sim.readField(fp.F3_FACE_B).ravel(), ora.By), (seed, frame)
        sim.density(); ora.deposit()
        fixed = sim.readField(fp.F3_RHO_FIXED)
        assert np.array_equal(fixed, ora.rho_fixed), (seed, frame)
    sim.destroy()


@pytest.mark.parametrize("form", ["1", "flat"])
@pytest.mark.parametrize("precision", ["fp32", "fp64"])
def test_chained_lattice_step_is_bit_identical(fp, eo, monkeypatch, precision, form):
    """The chained lattice step (round 4, the default of an undecomposed handle; FPIC_EM_CHAIN=0 / flat select the four sweeps /
    the form without LDS): the second B half step of a sub-step, the next sub-step's node centring and its first B half step
    as ONE sweep between two half-time arrays, B of the integer time formed only when somebody reads it.  Against the
    oracle's four-sweep cycle: particles, the integer current and both lattice fields bit for bit after every frame, with
    read-backs (which close the chain) at different points of it, an odd sub-step in between and the switch turned off half
    way; and the same for the form without LDS."""
    monkeypatch.setenv("FPIC_EM_CHAIN", form)
    rng = np.random.default_rng(44)
    shape, L, n = (12, 10, 16), (0.012, 0.010, 0.016), 6000
    dt = cfl_dt(shape, L, 0.5)
    spec = em_spec(shape, L, n, dt, macro_weight=1e6)
    dtype = np.float32 if precision == "fp32" else np.float64
    sim, ora = fp.makeCylindricalParticlePusher(spec, precision=precision), eo.OracleES3D(spec, dtype)
    pos, vel = rng.random((n, 3)) * L, rng.normal(0, 0.3, (n, 3))
    E, B = rng.normal(0, 1e4, shape + (3,)), rng.normal(0, 0.05, shape + (3,))
    for s in (sim, ora):
        s.set(position=pos, velocity=vel)
    sim.set(edge_E=E, face_B=B); ora.set_lattice(E=E, B=B)

    def same(tag, fields=True):
        got = sim.getParticles()
        assert same_bits(got["position"], ora.positions()) and same_bits(got["velocity"], ora.velocities()), tag
        if fields:
            assert np.array_equal(sim.readField(fp.F3_J_FIXED).ravel(), ora.J_fixed), tag
            assert same_bits(sim.readField(fp.F3_EDGE_E).ravel(), ora.Ey), tag
            assert same_bits(sim.readField(fp.F3_FACE_B).ravel(), ora.By), tag

    sim.step(); ora.step()                    # two sub-steps: one from the integer time, one chained
    same("after one frame")
    sim.step(3); ora.step(3)                  # six chained on
    same("after four frames, particles only", fields=False)      # (no read-back of B: the chain stays open)
    sim.substeps(1); ora.em_substep()
    same("after an odd sub-step")
    sim.step(2); ora.step(2)
    monkeypatch.setenv("FPIC_EM_CHAIN", "0")  # the switch goes off while the chain is open
    sim.step(); ora.step()
    same("after the switch went off")
    sim.destroy()


@pytest.mark.parametrize("precision", ["fp32", "fp64"])
@pytest.mark.parametrize("form", ["chained", "four sweeps"])
def test_chained_lattice_step_on_the_ranks_of_a_decomposition(fp, eo, monkeypatch, precision, form):
    """Round 4: the ranks of a full-EM decomposition run the chained lattice step too — each forms the half-time B of its slab
    AND of its halo planes itself, from the E halo it receives (one plane deeper above), so the halo copy of B is gone.
    Four ranks on a (24, 16, 48) lattice, two ghost planes, compared with ONE handle bit for bit: (1) six frames WITHOUT any
    read-back in between (the chain stays open across frames and migrations); (2) then B is read from rank 1 ALONE — it forms
    B of the integer time from what it holds and reopens by itself, the others stay open, nothing is exchanged for it — and
    two more frames; (3) a checkpoint-free close of every rank at the end.  FPIC_EM_CHAIN=0: the same with four sweeps and
    both halo copies."""
    monkeypatch.setenv("FPIC_EM_CHAIN", "1" if form == "chained" else "0")
    world, shape, ghost = 4, (24, 16, 48), 2
    rng = np.random.default_rng(77)
    n = 30000
    L = tuple(1e-3 * s for s in shape)
    spec = em_spec(shape, L, n, cfl_dt(shape, L), macro_weight=1e6)
    nzl = shape[2] // world
    pos, vel = rng.random((n, 3)) * L, rng.normal(0, 0.3, (n, 3))
    owner = np.floor(pos[:, 2] / L[2] * shape[2]).astype(int) // nzl
    order = np.argsort(owner, kind="stable")
    pos, vel, counts = pos[order], vel[order], np.bincount(owner, minlength=world)
    E, B = rng.normal(0, 1e4, shape + (3,)), rng.normal(0, 0.05, shape + (3,))
    one = fp.makeCylindricalParticlePusher(spec, precision=precision)
    one.set(position=pos, velocity=vel)
    one.set(edge_E=E, face_B=B)
    ranks = []
    for r in range(world):
        s = fp.makeCylindricalParticlePusher(spec, precision=precision)
        s.domainInit(r, world, ghost_planes=ghost, migrate_every=3)
        first = int(counts[:r].sum())
        s.domainSet(pos[first:first + counts[r]], vel[first:first + counts[r]], first_id=first)
        s.set(edge_E=E, face_B=B)
        ranks.append(s)
    group = fp.BoxGroup(ranks)

    def compare(tag, which_ranks=range(world), fields=(fp.F3_EDGE_E, fp.F3_FACE_B)):
        for which in fields:
            ref = one.readField(which).reshape(shape[2], -1)
            for r in which_ranks:
                got = ranks[r].readField(which).reshape(shape[2], -1)
                assert same_bits(got[r * nzl:(r + 1) * nzl], ref[r * nzl:(r + 1) * nzl]), (tag, which, r)

    def compare_particles(tag):
        parts = [s.domainGet() for s in ranks]
        ids = np.concatenate([p["ids"] for p in parts])
        assert np.array_equal(np.sort(ids), np.arange(n)), tag
        ref = one.getParticles()
        assert same_bits(np.concatenate([p["position"] for p in parts])[np.argsort(ids)], ref["position"]), tag
        assert same_bits(np.concatenate([p["velocity"] for p in parts])[np.argsort(ids)], ref["velocity"]), tag

    for frame in range(6):
        one.step(); group.step()
    compare_particles("six frames without a read-back")
    compare("rank 1 alone", which_ranks=[1])        # (closes rank 1's chain, and the one handle's)
    for frame in range(2):
        one.step(); group.step()
    compare_particles("after one rank had closed")
    compare("the end")
    jref = one.readField(fp.F3_J_FIXED).reshape(shape[2], -1)
    for r, s in enumerate(ranks):
        got = s.readField(fp.F3_J_FIXED).reshape(shape[2], -1)
        assert np.array_equal(got[r * nzl:(r + 1) * nzl], jref[r * nzl:(r + 1) * nzl]), r
    stats = [s.domainStats() for s in ranks]
    assert sum(s["migrated"] for s in stats) > 0 and all(s["lost"] == 0 for s in stats)
    for s in ranks + [one]:
        s.destroy()
