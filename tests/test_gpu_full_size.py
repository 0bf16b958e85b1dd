"""BASELINE.json configs[2], [3] and [4] AT FULL SIZE on one MI355X, asserted (not only timed): the frame they step is
the reference's `step(); density();` loop (empic.js:1436-1469, fusionsim.js:170-178) in the CART3D extension.

PARITY UNPINNED for these configurations (no reference counterpart; the oracle is the build's own definition,
oracle/es3d_oracle_impl.h).  What is asserted at full size is (1) what the domain offers independently of the size — the
exact integer total of the charge grid, the integer continuity equation of the Yee lattice at every node, Gauss's law at
rounding level, every particle inside the box, the total momentum — and (2) the oracle itself on a SAMPLE: the field is
read back, `fpic_substeps(1)` advances exactly one leap-frog sub-step, and oracle/es3d_oracle pushes every k-th particle
in that very field: positions, velocities and cells bit-exact.  Populations are generated on the device exactly as
bench.py does (c4_rank_particles); a 5e8-particle host array would take minutes of numpy.

Budget: the three tests together take about three minutes of the GPU suite's fifteen.
"""
import os
import sys

import numpy as np
import pytest

from helpers import ROOT, same_bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fp():
    import fusionpic
    fusionpic.load_library()
    return fusionpic


@pytest.fixture(scope="module")
def eo():
    import es3d_oracle
    return es3d_oracle


@pytest.fixture(scope="module")
def bench():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench as b
    return b


def exact_sum(a):
    """sum of an int64 array as a Python integer (5e8 particles x 2^42 does not fit 64 bits)"""
    a = np.ascontiguousarray(a).ravel()
    lo = int((a & np.int64((1 << 21) - 1)).sum(dtype=np.int64))
    hi = int((a >> np.int64(21)).sum(dtype=np.int64))
    return (hi << 21) + lo


def upload_generated(bench, sim, total, chunks, L, vth, species=0, mass_ratio=1.0):
    """`total` particles of a species in `chunks` pieces (z-slab r of `chunks` each), generated on the device; returns the
    total momentum the generator gave them (float64, per component)"""
    import torch
    assert total % chunks == 0
    share = total // chunks
    p0 = np.zeros(3)
    for r in range(chunks):
        pos, vel = bench.c4_rank_particles(r, chunks, species, share, L, vth, mass_ratio, 0)
        p0 += vel.sum(dim=0, dtype=torch.float64).cpu().numpy()
        sim.setRange(r * share, position=pos, velocity=vel, species=species)
        del pos, vel
    torch.cuda.empty_cache()
    return p0


def oracle_substep_on_a_sample(eo, spec, species_mass, species_charge, before, e4, idx, b0=(0.0, 0.0, 0.0)):
    """one electrostatic sub-step (es3d_push) of the particles `idx` of `before` (normalised positions, velocities as the
    library returned them) in the node field e4; returns positions, velocities, cells"""
    ora = eo.OracleES3D(dict(spec, count=len(idx), particle_mass=species_mass, particle_charge=species_charge), np.float32)
    sp = ora.species[0]
    for k, name in enumerate(("x", "y", "z")):
        getattr(sp, name)[:] = before["position"][idx, k]
    for k, name in enumerate(("vx", "vy", "vz")):
        getattr(sp, name)[:] = before["velocity"][idx, k]
    ora.E4[:] = np.ascontiguousarray(e4, dtype=np.float32).ravel()
    ora.B0[:] = b0
    ora.push()
    return ora.positions(), ora.velocities(), ora.cells()


def test_configs2_at_full_size(fp, eo, bench):
    """BASELINE configs[2]: 256^3 nodes, 5e8 electrons, fp32, Poisson solve every sub-step.  After 10 sub-steps (the first
    binning and at least one re-binning fused into the push): the charge grid sums to N 2^42 exactly, every particle lies
    in [0,1)^3, the total momentum moved by no more than the solve's rounding; then one more sub-step in the field read
    back beforehand, checked against the oracle on every 5000th particle bit for bit."""
    n, grid = 500_000_000, 256
    spec, L, vth = bench.es3d_scene(n, grid)
    sim = fp.makeCylindricalParticlePusher(spec)
    p0 = upload_generated(bench, sim, n, 10, L, vth)
    sim.precalc()
    sim.step(5)
    st = sim.stats()
    assert st["particle_updates"] == 10 * n and st["sort_passes"] >= 2, st    # first binning + a fused re-binning
    fixed = sim.readField(fp.F3_RHO_FIXED)
    assert exact_sum(fixed) == n * eo.FIXED_ONE
    before = sim.getParticles()
    pos = before["position"]
    assert float(pos.min()) >= 0.0 and float(pos.max()) < 1.0
    p1 = before["velocity"].sum(axis=0, dtype=np.float64)
    # the gather and the deposit are adjoint (same quantised weights): the field exerts no net force, up to the rounding
    # of 10 sub-steps of 5e8 single-precision kicks
    kick = 10 * 1e-6 * vth
    assert np.abs(p1 - p0).max() <= kick * np.sqrt(n), (p0, p1)
    e4 = sim.readField(fp.F3_E)
    sim.substeps(1)
    idx = np.arange(0, n, 5000)
    want_p, want_v, want_c = oracle_substep_on_a_sample(eo, spec, spec["particle_mass"], spec["particle_charge"], before, e4, idx)
    del before, pos
    after = sim.getParticles()
    assert same_bits(after["position"][idx], want_p)
    assert same_bits(after["velocity"][idx], want_v)
    assert np.array_equal(sim.getCells()[idx], want_c)
    assert exact_sum(sim.readField(fp.F3_RHO_FIXED)) == n * eo.FIXED_ONE
    assert sim.stats()["particle_updates"] == 11 * n
    sim.destroy()


def test_configs3_at_full_size(fp, eo, bench):
    """BASELINE configs[3]: 512^3 nodes, 1e9 electrons + 1e9 protons, fp32, on ONE handle (117 GB): with both species on
    it the charge grid sums to (N_e - N_p) 2^42 = 0 exactly (charge numbers +1 and -1 in units of the first species'
    charge), before and after 8 sub-steps; every proton lies inside the box; one more sub-step of the protons (the second
    species: its own mass, charge number and slab offsets beyond 2^32 bytes) in the field read back beforehand is the
    oracle's bit for bit on every 20000th one.  Then the decomposition at this grid: 8 in-process ranks against one handle
    on a population that can sit beside it (1e8 + 1e8), bit-identical on every rank's planes and for every particle, none
    lost."""
    import torch
    total, grid, world = 2_000_000_000, 512, 8
    spec, L, vth, mi, qi = bench.c4_scene(total, grid, world)
    ne = total // 2
    one = fp.makeCylindricalParticlePusher(dict(spec, count=ne))
    one.addSpecies(mi, qi, ne)
    upload_generated(bench, one, ne, 8, L, vth, species=0)
    upload_generated(bench, one, ne, 8, L, vth, species=1, mass_ratio=mi / spec["particle_mass"])
    one.precalc()
    assert exact_sum(one.readField(fp.F3_RHO_FIXED)) == 0            # Z = +1 and -1 in units of the first species' charge
    one.step(4)
    st = one.stats()
    assert st["particle_updates"] == 8 * total and st["sort_passes"] >= 1
    assert exact_sum(one.readField(fp.F3_RHO_FIXED)) == 0
    before = one.getParticles(species=1)
    assert float(before["position"].min()) >= 0.0 and float(before["position"].max()) < 1.0
    e4 = one.readField(fp.F3_E)
    one.substeps(1)
    idx = np.arange(0, ne, 20000)
    want_p, want_v, want_c = oracle_substep_on_a_sample(eo, spec, mi, qi, before, e4, idx)
    del before, e4
    after = one.getParticles(species=1)
    assert same_bits(after["position"][idx], want_p)
    assert same_bits(after["velocity"][idx], want_v)
    del after
    assert np.array_equal(one.getCells(species=1)[idx], want_c)
    one.destroy()
    torch.cuda.empty_cache()

    # the 8-slab decomposition on the same 512^3 grid (decomposed solve with the library's own transforms: the one handle's
    # field bit for bit; every rank keeps 77 of the 512 planes), 1e8 + 1e8 particles
    small = 200_000_000
    share = small // 2 // world
    spec2 = dict(spec, macro_weight=spec["macro_weight"] * total / small)
    one = fp.makeCylindricalParticlePusher(dict(spec2, count=share * world))
    one.addSpecies(mi, qi, share * world)
    ranks = []
    for r in range(world):
        s = fp.makeCylindricalParticlePusher(dict(spec2, count=int(share * 1.25)))
        s.addSpecies(mi, qi, int(share * 1.25))
        s.domainInit(r, world, ghost_planes=4, migrate_every=8, distributed_solve=True)
        ranks.append(s)
    for sp in range(2):
        for r in range(world):
            p, v = bench.c4_rank_particles(r, world, sp, share, L, vth, 1.0 if sp == 0 else mi / spec["particle_mass"], 0)
            one.setRange(r * share, position=p, velocity=v, species=sp)
            ranks[r].domainSet(p, v, first_id=r * share, species=sp)
            del p, v
    torch.cuda.empty_cache()
    group = fp.BoxGroup(ranks)
    one.precalc(); group.precalc()
    one.step(5); group.step(5)          # 10 sub-steps: one migration riding on the fused re-binning
    nzl = grid // world
    f1 = one.readField(fp.F3_RHO_FIXED).reshape(grid, -1)
    assert exact_sum(f1) == 0
    for r, s in enumerate(ranks):
        fr = s.readField(fp.F3_RHO_FIXED).reshape(grid, -1)
        assert np.array_equal(fr[r * nzl:(r + 1) * nzl], f1[r * nzl:(r + 1) * nzl]), r
    del f1, fr
    for sp in range(2):
        ref = one.getParticles(species=sp)
        parts = [s.domainGet(species=sp) for s in ranks]
        ids = np.concatenate([p["ids"] for p in parts])
        order = np.argsort(ids)
        assert len(ids) == share * world and np.array_equal(ids[order], np.arange(share * world, dtype=np.uint32)), sp
        assert same_bits(np.concatenate([p["position"] for p in parts])[order], ref["position"]), sp
        assert same_bits(np.concatenate([p["velocity"] for p in parts])[order], ref["velocity"]), sp
        del ref, parts
    stats = [s.domainStats() for s in ranks]
    assert sum(s["migrated"] for s in stats) > 0 and all(s["lost"] == 0 for s in stats), stats
    for s in ranks + [one]:
        s.destroy()
    torch.cuda.empty_cache()


def oracle_em_substep_on_a_sample(eo, spec, before, edge_e, face_b):
    """one full-EM sub-step's push (em_nodes + em_push of oracle/es3d_oracle_impl.h) of the sampled particles `before`
    (normalised positions, velocities as the library returned them) in the lattice fields read back beforehand"""
    count = before["position"].shape[0]
    ora = eo.OracleES3D(dict(spec, count=count), np.float64)
    sp = ora.species[0]
    for k, name in enumerate(("x", "y", "z")):
        getattr(sp, name)[:] = before["position"][:, k]
    for k, name in enumerate(("vx", "vy", "vz")):
        getattr(sp, name)[:] = before["velocity"][:, k]
    ora.Ey[:] = edge_e.ravel()
    ora.By[:] = face_b.ravel()
    ora.em_substep()      # nodes, push, current, field update: the particles are what is compared
    return ora.positions(0), ora.velocities(0), ora.cells(0)


def test_configs4_at_full_size(fp, eo, bench):
    """BASELINE configs[4] AS STATED on one GPU: 512^3 Yee lattice, fp64, 2e9 electrons (233 GB of the card's 288).  Over
    one sub-step the integer continuity equation holds at EVERY node: 96 (rho_fixed(n+1) - rho_fixed(n)) + div J_fixed = 0;
    Gauss's law on the lattice stays at rounding level from precalc() through the steps; and the oracle itself on a
    sample: the lattice fields are read back, every 20 000th particle is read with the ranged read-back
    (fpic_get_particles_range: 2e9 fp64 particles would be 96 GB of host arrays), fpic_substeps(1) advances one sub-step,
    and oracle/es3d_oracle's em_push moves the sample in that very field: positions, velocities, cells bit for bit."""
    import torch
    n, grid = 2_000_000_000, 512
    c, eps0, me, qe, vth, wp = 2.998e8, 8.8541878128e-12, 9.109e-31, -1.602e-19, 1e-3, 1e10
    dx = vth * c / wp
    L = grid * dx
    spec = dict(radius=L, length_y=L, height=L, nr=grid, ny=grid, nz=grid, dt=0.5 * dx / (c * 3 ** 0.5), nparticles=0, count=n, particle_mass=me,
                particle_charge=qe, geometry="cart3d", solver="yee", macro_weight=wp ** 2 * eps0 * me / qe ** 2 * L ** 3 / n, precision="fp64")
    sim = fp.makeCylindricalParticlePusher(spec)
    upload_generated(bench, sim, n, 32, L, vth)
    sim.precalc()
    shape = (grid, grid, grid)
    dv = (L / grid) ** 3
    rho_scale = qe * spec["macro_weight"] / (eo.FIXED_ONE * dv)

    def gauss():
        sim.density()
        rho = sim.readField(fp.F3_RHO_FIXED).reshape(shape).astype(np.float64) * rho_scale
        scale = np.abs(rho).max() / eps0
        rho -= rho.mean()
        rho /= eps0
        e = sim.readField(fp.F3_EDGE_E, np.float64).reshape(grid, grid, grid, 4)
        for a, axis in ((0, 2), (1, 1), (2, 0)):
            comp = e[..., a]
            rho -= (comp - np.roll(comp, 1, axis=axis)) / dx
        return np.abs(rho).max(), scale

    res, scale = gauss()
    assert res <= 1e-9 * scale, (res, scale)
    sim.step(2)
    sim.density()
    r0 = sim.readField(fp.F3_RHO_FIXED).copy()
    assert exact_sum(r0) == n * eo.FIXED_ONE
    # the sample and the fields it is about to be pushed in
    stride = 20000
    count = n // stride
    before = sim.getRange(0, count, stride)
    assert float(before["position"].min()) >= 0.0 and float(before["position"].max()) < 1.0
    edge_e, face_b = sim.readField(fp.F3_EDGE_E), sim.readField(fp.F3_FACE_B)
    sim.substeps(1)
    after = sim.getRange(0, count, stride, cells=True)
    J = sim.readField(fp.F3_J_FIXED).reshape(grid, grid, grid, 3)
    sim.density()
    r1 = sim.readField(fp.F3_RHO_FIXED)
    acc = (r1 - r0).reshape(shape)
    acc *= 96
    del r0, r1
    for a, axis in ((0, 2), (1, 1), (2, 0)):
        comp = np.ascontiguousarray(J[..., a])
        acc += comp
        acc -= np.roll(comp, 1, axis=axis)
    assert not acc.any()
    assert int(np.abs(J).max()) > 0
    del acc, J
    res, scale = gauss()
    assert res <= 1e-8 * scale, (res, scale)
    st = sim.stats()
    assert st["particle_updates"] == 5 * n
    sim.destroy()
    torch.cuda.empty_cache()
    want_p, want_v, want_c = oracle_em_substep_on_a_sample(eo, spec, before, edge_e, face_b)
    assert same_bits(after["position"], want_p)
    assert same_bits(after["velocity"], want_v)
    assert np.array_equal(after["cells"], want_c)
    assert not same_bits(after["position"], before["position"])


def test_configs4_decomposition_at_its_lattice(fp, eo, bench):
    """The full-EM decomposition at configs[4]'s lattice (until now asserted at 64 x 64 x 128): 512^3
    Yee lattice, fp64, 8 in-process ranks of 64 planes against ONE handle on a population that can sit beside it (2e8
    electrons): after 3 frames (6 sub-steps, one migration) the integer current grid and both lattice fields on every
    rank's planes and every particle are bit-identical to the one handle's, none lost.  The ranks take the decomposed solve
    for their start field and keep their slab and halo planes only, as a run of that size would (with whole-grid arrays
    eight ranks' two half-time arrays of the chained lattice step no longer fit beside the one handle: 69 GB)."""
    import torch
    n, grid, world = 200_000_000, 512, 8
    c, eps0, me, qe, vth, wp = 2.998e8, 8.8541878128e-12, 9.109e-31, -1.602e-19, 1e-3, 1e10
    dx = vth * c / wp
    L = grid * dx
    spec = dict(radius=L, length_y=L, height=L, nr=grid, ny=grid, nz=grid, dt=0.5 * dx / (c * 3 ** 0.5), nparticles=0, count=n, particle_mass=me,
                particle_charge=qe, geometry="cart3d", solver="yee", macro_weight=wp ** 2 * eps0 * me / qe ** 2 * L ** 3 / n, precision="fp64")
    share = n // world
    one = fp.makeCylindricalParticlePusher(spec)
    ranks = []
    for r in range(world):
        s = fp.makeCylindricalParticlePusher(dict(spec, count=int(share * 1.25)))
        s.domainInit(r, world, ghost_planes=2, migrate_every=4, distributed_solve=True)   # (a rank keeps 73 of the 512 planes)
        ranks.append(s)
    for r in range(world):
        p, v = bench.c4_rank_particles(r, world, 0, share, L, vth, 1.0, 0)
        one.setRange(r * share, position=p, velocity=v)
        ranks[r].domainSet(p, v, first_id=r * share)
        del p, v
    torch.cuda.empty_cache()
    group = fp.BoxGroup(ranks)
    one.precalc(); group.precalc()
    one.step(3); group.step(3)
    nzl = grid // world
    for which in (fp.F3_J_FIXED, fp.F3_EDGE_E, fp.F3_FACE_B):
        ref = one.readField(which).reshape(grid, -1)
        for r, s in enumerate(ranks):
            got = s.readField(which).reshape(grid, -1)[r * nzl:(r + 1) * nzl]
            assert (np.array_equal if which == fp.F3_J_FIXED else same_bits)(got, ref[r * nzl:(r + 1) * nzl]), (which, r)
            del got
        del ref
    ref = one.getParticles()
    parts = [s.domainGet() for s in ranks]
    ids = np.concatenate([p["ids"] for p in parts])
    order = np.argsort(ids)
    assert len(ids) == n and np.array_equal(ids[order], np.arange(n, dtype=np.uint32))
    assert same_bits(np.concatenate([p["position"] for p in parts])[order], ref["position"])
    assert same_bits(np.concatenate([p["velocity"] for p in parts])[order], ref["velocity"])
    stats = [s.domainStats() for s in ranks]
    assert sum(s["migrated"] for s in stats) > 0 and all(s["lost"] == 0 for s in stats), stats
    for s in ranks + [one]:
        s.destroy()
    torch.cuda.empty_cache()


def test_ranged_and_sampled_read_back(fp):
    """fpic_get_particles_range / fpic_get_cells_range: every range and stride of a binned, stepped box equals the same
    slice of the whole read-back (the caller's order, whatever the bins did); out-of-range requests are refused."""
    rng = np.random.default_rng(3)
    n, shape, L = 100_003, (32, 32, 16), (0.02, 0.02, 0.01)
    spec = dict(radius=L[0], length_y=L[1], height=L[2], nr=shape[0], ny=shape[1], nz=shape[2], dt=2e-11, nparticles=0, count=n,
                particle_mass=9.109e-31, particle_charge=-1.602e-19, geometry="cart3d", solver="poisson_fft", macro_weight=1e4)
    for precision in ("fp32", "fp64"):
        sim = fp.makeCylindricalParticlePusher(spec, precision=precision)
        sim.set(position=rng.random((n, 3)) * L, velocity=rng.normal(0, 0.05, (n, 3)))
        sim.precalc()
        sim.step(2)
        whole, cells = sim.getParticles(), sim.getCells()
        for first, count, stride in ((0, n, 1), (5, 1000, 1), (17, 3000, 31), (n - 1, 1, 1), (0, (n + 19999) // 20000, 20000), (99_000, 0, 1)):
            got = sim.getRange(first, count, stride, cells=True)
            idx = first + stride * np.arange(count)
            assert same_bits(got["position"], whole["position"][idx]) and same_bits(got["velocity"], whole["velocity"][idx])
            assert np.array_equal(got["cells"], cells[idx])
        other = np.float32 if precision == "fp64" else np.float64
        got = sim.getRange(10, 50, 7, dtype=other)
        assert np.array_equal(got["position"], whole["position"][10 + 7 * np.arange(50)].astype(other))
        for bad in ((n, 1, 1), (0, n + 1, 1), (0, 2, n), (0, 1, 0)):
            with pytest.raises(fp.FusionPicError):
                sim.getRange(*bad)
        sim.destroy()


# ------------------------------------------------------------------------------------------- the solve where it runs

EPS0 = 8.8541878128e-12


def numpy_poisson(rho, L):
    """phi of the 3-point-Laplacian Poisson problem in double with numpy's FFT: phi_hat = rho_hat / (eps0 K^2),
    K^2 = sum_axis (2/d sin(pi m/n))^2, mean mode 0 (the definition in oracle/es3d_oracle_impl.h, not its code)."""
    nz, ny, nx = rho.shape
    hat = np.fft.rfftn(rho)
    k2 = 0.0
    for axis, (n, length, half) in enumerate(((nz, L[2], False), (ny, L[1], False), (nx, L[0], True))):
        m = np.arange(n // 2 + 1) if half else np.arange(n)
        term = (2.0 * n / length * np.sin(np.pi * m / n)) ** 2
        shape = [1, 1, 1]
        shape[axis] = term.size
        k2 = k2 + term.reshape(shape)
    k2[0, 0, 0] = 1.0
    hat /= EPS0 * k2
    hat[0, 0, 0] = 0.0
    return np.fft.irfftn(hat, s=rho.shape, axes=(0, 1, 2))


@pytest.mark.parametrize("n,precision", [(256, "fp32"), (256, "fp64"), (512, "fp32")])
def test_poisson_solve_against_numpy_at_the_sizes_it_runs_at(fp, n, precision):
    """the library's own FFT passes (csrc/fes_fft.hpp: 256- and 512-point Stockham sweeps, whose
    launch shapes differ from the <= 64-point cases the oracle comparison reaches) asserted DIRECTLY at the grid sizes of
    BASELINE configs[2] (256^3) and configs[3] / [4] (512^3), against an independent solve: numpy.fft.rfftn in double of
    the library's own charge density, divided by the eigenvalues of the 3-point Laplacian.  phi within 2e-5 (fp32) /
    1e-10 (fp64) of the largest potential; E = -grad phi by central differences of the library's own phi."""
    rng = np.random.default_rng(n)
    count = 100000
    L = (0.7, 1.3, 0.9)
    spec = dict(radius=L[0], length_y=L[1], height=L[2], nr=n, ny=n, nz=n, dt=1e-10, nparticles=0, count=count,
                particle_mass=9.109e-31, particle_charge=-1.602e-19, geometry="cart3d", solver="poisson_fft", macro_weight=1e9)
    sim = fp.makeCylindricalParticlePusher(spec, precision=precision)
    # a lumpy cloud: every wavelength from the box down to a few cells carries charge
    centres = rng.random((40, 3))
    pos = (centres[rng.integers(0, 40, count)] + rng.normal(0, 0.03, (count, 3)) * rng.random((count, 1))) % 1.0 * L
    sim.set(position=pos, velocity=np.zeros((count, 3)))
    sim.precalc()
    rho = sim.readField(fp.F3_RHO, np.float64).reshape(n, n, n)          # [k][j][i]
    fixed = sim.readField(fp.F3_RHO_FIXED)
    assert exact_sum(fixed) == count * (1 << 42)
    want = numpy_poisson(rho, L)
    del fixed
    phi = sim.readField(fp.F3_PHI, np.float64).reshape(n, n, n)
    tol = 2e-5 if precision == "fp32" else 1e-10
    top = np.abs(want).max()
    assert top > 0
    assert np.abs(phi - want).max() <= tol * top, np.abs(phi - want).max() / top
    del want, rho
    e4 = sim.readField(fp.F3_E, np.float64).reshape(n, n, n, 4)
    assert same_bits(e4[..., 3], phi)
    worst, scale = 0.0, 0.0
    for comp, (axis, length) in enumerate(((2, L[0]), (1, L[1]), (0, L[2]))):
        grad = -(np.roll(phi, -1, axis=axis) - np.roll(phi, 1, axis=axis)) * (n / (2.0 * length))
        worst = max(worst, float(np.abs(e4[..., comp] - grad).max()))
        scale = max(scale, float(np.abs(grad).max()))
        del grad
    # the gradient is formed in T from phi in T: its rounding is eps(T) * |phi| * n / (2 L) per node
    eps = np.finfo(np.float32 if precision == "fp32" else np.float64).eps
    assert worst <= 4 * eps * top * n / (2 * min(L)) + 1e-6 * scale, (worst, scale)
    sim.destroy()
