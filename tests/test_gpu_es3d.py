"""GPU parity tests of the CART3D electrostatic extension (spec.geometry = 'cart3d') through the
C ABI, against the build's own CPU oracle (oracle/es3d_oracle.c) and analytic known answers.

PARITY UNPINNED — the reference has no self-consistent mode: what is
compared here is the HIP path with the oracle that defines the mode.  Bar: particles, cell
indices and the fixed-point charge grid bit-exact for a given field; the FFT solve within 2e-5
(fp32) / 1e-10 (fp64) of the oracle's double-precision solve; Poisson single mode to 1e-6; cold
plasma oscillation within 1 % of omega_p; total charge exact.
"""
import numpy as np
import pytest

from helpers import same_bits

pytestmark = pytest.mark.gpu

ME, QE = 9.109e-31, -1.602e-19
MP = 1.67e-27


@pytest.fixture(scope="module")
def fp():
    import fusionpic
    fusionpic.load_library()
    return fusionpic


@pytest.fixture(scope="module")
def eo():
    import es3d_oracle
    return es3d_oracle


def box_spec(n=(16, 16, 16), L=(1.0, 1.0, 1.0), count=1000, dt=1e-10, mass=ME, charge=QE, **kw):
    s = dict(radius=L[0], length_y=L[1], height=L[2], nr=n[0], ny=n[1], nz=n[2], dt=dt, nparticles=0, count=count,
             particle_mass=mass, particle_charge=charge, geometry="cart3d", solver="poisson_fft", macro_weight=1.0)
    s.update(kw)
    return s


def make_pair(fp, eo, spec, precision):
    dtype = np.float32 if precision == "fp32" else np.float64
    return fp.makeCylindricalParticlePusher(spec, precision=precision), eo.OracleES3D(spec, dtype)


def assert_same_particles(sim, ora, species=0, what=""):
    got = sim.getParticles(species=species)
    assert np.array_equal(sim.getCells(species=species), ora.cells(species)), what
    assert same_bits(got["position"], ora.positions(species)), what
    assert same_bits(got["velocity"], ora.velocities(species)), what


@pytest.mark.parametrize("precision", ["fp32", "fp64"])
@pytest.mark.parametrize("shape,n,with_b,sort_interval", [((16, 16, 8), 4000, False, 0), ((40, 24, 20), 20001, True, 1), ((12, 10, 6), 999, True, 2),
                                                          ((64, 32, 32), 50003, False, 1), ((48, 48, 24), 30002, True, 3),
                                                          ((32, 16, 8), 7, False, 1)])
def test_push_and_deposit_bit_exact_in_a_given_field(fp, eo, precision, shape, n, with_b, sort_interval):
    """solver 'none': a random static E, uniform B; every sub-step's particles, cells and charge grid are
    bit-identical to the oracle's (flat path at precalc(), tiled path afterwards, re-binning in between;
    grids smaller than, equal to and not a multiple of the 16x16x8 tile; counts not a multiple of 4;
    sort_interval 1..3: the re-binning fused into the push runs every, every second, every third sub-step)."""
    rng = np.random.default_rng(n)
    L = (0.02, 0.015, 0.01)
    spec = box_spec(shape, L, count=n, dt=2e-11, solver="none", macro_weight=2e4)
    dtype = np.float32 if precision == "fp32" else np.float64
    sim, ora = fp.makeCylindricalParticlePusher(spec, precision=precision, sort_interval=sort_interval), eo.OracleES3D(spec, dtype)
    pos = rng.random((n, 3)) * L
    pos[:6] = [[0, 0, 0], [L[0], L[1], L[2]], [L[0] * (1 - 1e-9), 0, 0], [-0.001, 0.02, 0.0101], [L[0] / shape[0], L[1] / shape[1], 0],
               [L[0] * 0.5, L[1] * 0.999999, L[2] * 0.5]]
    # a third of the particles cross one to two cells per sub-step: exercises the wrap and the out-of-window path
    vel = rng.normal(0, 0.01, (n, 3)) + rng.normal(0, 0.3, (n, 3)) * (rng.random((n, 1)) < 0.3)
    E = rng.normal(0, 3e4, shape + (3,))
    for s in (sim, ora):
        s.set(position=pos, velocity=vel, E=E)
    if with_b:
        sim.addB(0.3, -0.2, 0.9); ora.add_b(0.3, -0.2, 0.9)
    sim.precalc(); ora.precalc()
    assert np.array_equal(sim.readField(fp.F3_RHO_FIXED), ora.rho_fixed)
    assert_same_particles(sim, ora, what="upload")
    for frame in range(3):
        sim.step(); ora.step()
        assert_same_particles(sim, ora, what="frame %d" % frame)
        fixed = sim.readField(fp.F3_RHO_FIXED)
        assert np.array_equal(fixed, ora.rho_fixed), frame
        assert int(fixed.sum()) == n * eo.FIXED_ONE
    assert same_bits(sim.readField(fp.F3_RHO), ora.rho)
    st = sim.stats()
    assert st["particle_updates"] == 6 * n and st["sort_passes"] >= 1
    if sort_interval:
        assert st["sort_passes"] == 1 + 5 // sort_interval
    sim.destroy()


@pytest.mark.parametrize("precision", ["fp32", "fp64"])
@pytest.mark.parametrize("shape", [(16, 16, 16), (32, 8, 12), (24, 20, 16), (64, 64, 64)])
def test_poisson_solve_matches_oracle_and_single_mode(fp, eo, precision, shape):
    """rho from a random particle cloud: phi and E within tolerance of the oracle's double-precision solve;
    and the discrete operator's eigenvector: rho = cos(k.x) gives phi = rho / (eps0 K^2) to 1e-6"""
    rng = np.random.default_rng(sum(shape))
    n = 30000
    L = (0.7, 1.3, 0.9)
    spec = box_spec(shape, L, count=n, macro_weight=1e9)
    sim, ora = make_pair(fp, eo, spec, precision)
    pos = rng.random((n, 3)) * L
    for s in (sim, ora):
        s.set(position=pos, velocity=np.zeros((n, 3)))
    sim.precalc(); ora.precalc()
    assert np.array_equal(sim.readField(fp.F3_RHO_FIXED), ora.rho_fixed)
    assert same_bits(sim.readField(fp.F3_RHO), ora.rho)
    tol = 2e-5 if precision == "fp32" else 1e-10
    phi, want = sim.readField(fp.F3_PHI, np.float64), ora.phi.astype(np.float64)
    assert np.abs(phi - want).max() <= tol * np.abs(want).max()
    e4, w4 = sim.readField(fp.F3_E, np.float64), ora.E4.reshape(-1, 4).astype(np.float64)
    assert np.abs(e4[:, :3] - w4[:, :3]).max() <= 20 * tol * np.abs(w4[:, :3]).max()
    assert same_bits(e4[:, 3], phi)
    sim.destroy()


@pytest.mark.parametrize("precision", ["fp32", "fp64"])
def test_self_consistent_steps_track_the_oracle(fp, eo, precision):
    """a warm two-species plasma, five frames of push+deposit+solve: the charge grid after the first
    sub-step is bit-identical (same E to the last bit is not required for that: cells only), total
    charge is exact every frame, particles stay within the solve's tolerance of the oracle's"""
    rng = np.random.default_rng(77)
    n = 40000
    shape, L = (32, 16, 24), (0.032, 0.016, 0.024)
    dens = 1e15
    spec = box_spec(shape, L, count=n, dt=5e-12, macro_weight=dens * np.prod(L) / n)
    sim, ora = make_pair(fp, eo, spec, precision)
    ions_s = sim.addSpecies(MP, -QE, n // 2)
    ions_o = ora.add_species(MP, -QE, n // 2)
    assert ions_s == ions_o == 1
    pe, ve = rng.random((n, 3)) * L, rng.normal(0, 2e-3, (n, 3))
    pi, vi = rng.random((n // 2, 3)) * L, rng.normal(0, 5e-5, (n // 2, 3))
    for s in (sim, ora):
        s.set(position=pe, velocity=ve)
        s.set(position=pi, velocity=vi, species=1)
    sim.addBZ(0.05); ora.add_bz(0.05)
    sim.precalc(); ora.precalc()
    assert np.array_equal(sim.readField(fp.F3_RHO_FIXED), ora.rho_fixed)
    total = (n - n // 2) * eo.FIXED_ONE
    tol = 1e-4 if precision == "fp32" else 1e-9
    for frame in range(5):
        sim.step(); ora.step()
        fixed = sim.readField(fp.F3_RHO_FIXED)
        assert int(fixed.sum()) == total
        for sp in (0, 1):
            got = sim.getParticles(np.float64, species=sp)
            d = np.abs(got["position"] - ora.positions(sp)); d = np.minimum(d, 1 - d)
            assert d.max() <= tol, (frame, sp)
            vs = np.abs(ora.velocities(sp)).max()
            assert np.abs(got["velocity"] - ora.velocities(sp)).max() <= 20 * tol * vs, (frame, sp)
            assert np.mean(sim.getCells(species=sp) == ora.cells(sp)) > 0.999
    e4, w4 = sim.readField(fp.F3_E, np.float64), ora.E4.reshape(-1, 4).astype(np.float64)
    assert np.abs(e4[:, :3] - w4[:, :3]).max() <= 100 * tol * np.abs(w4[:, :3]).max()
    sim.destroy()


def test_cold_plasma_oscillation_on_the_gpu(fp, eo):
    """SURVEY section 7 known answer: field energy oscillates at 2 omega_p; grid dispersion omega_p cos(k dx/2)"""
    nx, L, per_cell, dt, wp_dt = 32, 1.0, 4, 1e-10, 0.05
    n = nx * per_cell * 4 * 4
    wp = wp_dt / dt
    density = wp ** 2 * eo.EPS0 * ME / QE ** 2
    spec = box_spec((nx, 4, 4), (L, L / 8, L / 8), count=n, dt=dt, macro_weight=density * L * (L / 8) ** 2 / n)
    sim = fp.makeCylindricalParticlePusher(spec)
    xs = (np.arange(nx * per_cell) + 0.5) / (nx * per_cell) * L
    ys = (np.arange(4) + 0.5) / 4 * (L / 8)
    X, Y, Z = np.meshgrid(xs, ys, ys, indexing="ij")
    pos = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    vel = np.zeros_like(pos)
    vel[:, 0] = 2e-3 * np.sin(2 * np.pi / L * pos[:, 0])
    sim.set(position=pos, velocity=vel)
    sim.precalc()
    dv = (L / nx) * (L / 32) ** 2
    energy = []
    for _ in range(int(round(2.2 * 2 * np.pi / wp_dt)) // 2):
        sim.step()
        e4 = sim.readField(fp.F3_E, np.float64)
        energy.append(0.5 * eo.EPS0 * float((e4[:, :3] ** 2).sum()) * dv)
    energy = np.array(energy)
    t = (np.arange(len(energy)) + 1) * 2 * dt
    peaks = [i for i in range(1, len(energy) - 1) if energy[i] > energy[i - 1] and energy[i] >= energy[i + 1]]
    assert len(peaks) >= 3

    def vertex(i):
        y0, y1, y2 = energy[i - 1], energy[i], energy[i + 1]
        return t[i] + 0.5 * (y0 - y2) / (y0 - 2 * y1 + y2) * (t[1] - t[0])
    omega = np.pi / np.mean(np.diff([vertex(i) for i in peaks]))
    assert abs(omega / wp - 1) < 0.01
    assert abs(omega / (wp * np.cos(np.pi / nx)) - 1) < 0.002
    sim.destroy()


def test_box_error_paths(fp):
    import ctypes
    lib = fp.load_library()
    spec = box_spec((8, 8, 8), count=10)
    sim = fp.makeCylindricalParticlePusher(spec)
    h = sim._h
    msg = lambda: lib.fpic_last_error(h).decode()
    f32 = np.zeros(8 * 8 * 8 * 3, dtype=np.float32)
    assert lib.fpic_step(h, 1) == -5 and "precalc" in msg()
    assert lib.fpic_set_grid(h, 0, f32.ctypes.data, 8, 8, 3, 0) == -5 and "CART3D" in msg()
    assert lib.fpic_read_grid(h, 0, f32.ctypes.data, 0) == -5
    assert lib.fpic_add_current_loop(h, 0.1, 0.1, 1.0) == -5
    assert lib.fpic_set_random_state(h, None, None) == -5
    assert lib.fpic_load_checkpoint(h, b"/nonexistent/x.ckp") == -5 and "cannot open" in msg()   # (a box does checkpoint: see below)
    assert lib.fpic_set_field3(h, 0, f32.ctypes.data, 8, 8, 4, 0) == -1 and ".grid" in msg()
    assert lib.fpic_set_field3(h, 1, f32.ctypes.data, 8, 8, 8, 0) == -1 and ".which" in msg()
    assert lib.fpic_read_field3(h, 9, f32.ctypes.data, 0) == -1
    assert lib.fpic_set_particles_of(h, 3, f32.ctypes.data, None, 10, 0) == -1 and ".species" in msg()
    assert lib.fpic_set_particles_of(h, 0, f32.ctypes.data, None, 11, 0) == -1 and "expected 10 particles" in msg()
    idx = ctypes.c_int()
    assert lib.fpic_add_species(h, 1.0, 0.5 * QE, 5, ctypes.byref(idx)) == -1 and ".charge" in msg()
    assert lib.fpic_add_species(h, -1.0, QE, 5, ctypes.byref(idx)) == -1 and ".mass" in msg()
    # box-only calls on the reference's pusher
    rz = fp.makeCylindricalParticlePusher(dict(radius=1.0, height=1.0, nr=8, nz=8, dt=1e-9, nparticles=3, particle_mass=MP,
                                               particle_charge=-QE))
    assert lib.fpic_add_b(rz._h, 0.0, 0.0, 1.0) == -5 and b"CART3D" in lib.fpic_last_error(rz._h)
    assert lib.fpic_read_field3(rz._h, 0, f32.ctypes.data, 0) == -5
    # bad box specs
    bad = dict(spec, ny=0)
    with pytest.raises(fp.FusionPicError) as e:
        fp.makeCylindricalParticlePusher(bad)
    assert ".ny" in str(e.value)
    with pytest.raises(fp.FusionPicError) as e:
        fp.makeCylindricalParticlePusher(dict(spec, length_y=-1.0))
    assert ".length_y" in str(e.value)
    # the handle still works
    sim.set(position=np.random.default_rng(0).random((10, 3)), velocity=np.zeros((10, 3)))
    sim.precalc(); sim.step(2); sim.density()
    got = sim.getParticles()
    assert np.isfinite(got["position"]).all() and ((got["position"] >= 0) & (got["position"] < 1)).all()
    assert lib.fpic_get_particles(h, None, None, None, None, 0) == 0
    sim.destroy(); rz.destroy()


def test_large_box_properties(fp, eo):
    """128^3 nodes, 2e7 particles (the shape of BASELINE configs[2] at 1/25 of its size): exact total charge,
    momentum conserved, every particle inside the box, re-binning kept the caller's order."""
    n = 20_000_000
    shape, L = (128, 128, 128), (0.128, 0.128, 0.128)
    dens, dt = 1e15, 2e-12
    spec = box_spec(shape, L, count=n, dt=dt, macro_weight=dens * np.prod(L) / n)
    sim = fp.makeCylindricalParticlePusher(spec)
    rng = np.random.Generator(np.random.Philox(0x5EEDF051))
    pos = rng.random((n, 3), dtype=np.float32) * np.float32(L[0])
    vel = rng.standard_normal((n, 3), dtype=np.float32) * np.float32(1e-3)
    sim.set(position=pos, velocity=vel)
    sim.precalc()
    p0 = vel.astype(np.float64).sum(axis=0)
    sim.step(4)
    fixed = sim.readField(fp.F3_RHO_FIXED)
    assert int(fixed.astype(object).sum()) == n * eo.FIXED_ONE
    got = sim.getParticles()
    assert ((got["position"] >= 0) & (got["position"] < 1)).all()
    p1 = got["velocity"].astype(np.float64).sum(axis=0)
    dv = np.abs(got["velocity"].astype(np.float64) - vel).max()
    assert dv > 0 and np.abs(p1 - p0).max() <= 1e-3 * dv * np.sqrt(n)
    # the caller's order survived the re-binnings: a slow particle is still near where it started
    slow = np.argsort(np.abs(vel).max(axis=1))[:1000]
    d = np.abs(got["position"][slow] - pos[slow] / np.float32(L[0])); d = np.minimum(d, 1 - d)
    assert d.max() < 0.05
    st = sim.stats()
    assert st["sort_passes"] >= 1 and st["particle_updates"] == 8 * n
    sim.destroy()


def test_box_through_the_javascript_host(fp, eo, tmp_path):
    """Node -> empic_native.js (spec.geometry 'cart3d') -> N-API addon -> libfusionpic.so: two species, a field
    solve per sub-step; charge grid bit-identical to the oracle's after precalc(), particles within the solve's
    tolerance after three frames; the library's communicator with a world of one."""
    import base64
    import json
    import os
    import shutil
    import subprocess
    from helpers import ROOT
    node = shutil.which("node")
    if node is None:
        pytest.skip("node is not installed on this box")
    rng = np.random.default_rng(12)
    n, ni = 3000, 1200
    shape, L = (16, 12, 8), (0.016, 0.012, 0.008)
    spec = box_spec(shape, L, count=n, dt=5e-12, macro_weight=1e15 * np.prod(L) / n)
    pe, ve = rng.random((n, 3)) * L, rng.normal(0, 2e-3, (n, 3))
    pi, vi = rng.random((ni, 3)) * L, rng.normal(0, 5e-5, (ni, 3))
    (tmp_path / "in.json").write_text(json.dumps(dict(spec=spec, pe=pe.tolist(), ve=ve.tolist(), pi=pi.tolist(), vi=vi.tolist(), mp=MP, qi=-QE)))
    script = r"""
const fs = require('fs');
const empic = require(process.argv[1]);
const inp = JSON.parse(fs.readFileSync(process.argv[2]));
const sim = empic.makeCylindricalParticlePusher(inp.spec);
const ions = sim.addSpecies(inp.mp, inp.qi, inp.pi.length);
sim.set({position: inp.pe, velocity: inp.ve});
sim.set({position: inp.pi, velocity: inp.vi}, ions);
sim.addBZ(0.05);
sim.commInit(empic.commUniqueId(), 0, 1);
sim.precalc();
const b64 = a => Buffer.from(a.buffer, a.byteOffset, a.byteLength).toString('base64');
const fixed0 = b64(sim.readField('rho_fixed'));
for (let frame = 0; frame < 3; frame++) { sim.step(); sim.density(); }
const e = sim.getParticles(), i = sim.getParticles(null, ions);
const some = sim.getRange(7, 100, null, ions, 11);            // ions 7, 18, 29, ...: the sampled read-back
let err = 'none';
try { sim.readField('E', new Float32Array(7)); } catch (x) { err = x.constructor.name; }
console.log(JSON.stringify({ions: ions, fixed0: fixed0, fixed: b64(sim.readField('rho_fixed')), pe: b64(e.position), ve: b64(e.velocity),
  pi: b64(i.position), some: b64(some.position), comm: sim.commInfo(), cells: b64(sim.getCells()), E: b64(sim.readField('E')), err: err, updates: sim.stats().particle_updates}));
sim.destroy();
"""
    shim = os.path.join(ROOT, "fusion-sim_amd", "js", "empic_native.js")
    raw = subprocess.check_output([node, "-e", script, shim, str(tmp_path / "in.json")])
    out = json.loads(raw.decode().strip().splitlines()[-1])
    dec = lambda k, dt: np.frombuffer(base64.b64decode(out[k]), dtype=dt)
    ora = eo.OracleES3D(spec, np.float32)
    assert ora.add_species(MP, -QE, ni) == out["ions"] == 1
    ora.set(position=pe, velocity=ve); ora.set(position=pi, velocity=vi, species=1)
    ora.add_bz(0.05)
    ora.precalc()
    assert np.array_equal(dec("fixed0", np.int64), ora.rho_fixed)
    ora.step(3)
    assert int(dec("fixed", np.int64).sum()) == (n - ni) * eo.FIXED_ONE
    for key, want in (("pe", ora.positions(0)), ("pi", ora.positions(1))):
        d = np.abs(dec(key, np.float32).reshape(-1, 3).astype(np.float64) - want); d = np.minimum(d, 1 - d)
        assert d.max() <= 1e-4
    assert np.mean(dec("cells", np.int32) == ora.cells(0)) > 0.995
    assert out["err"] == "RangeError" and out["updates"] == 6 * (n + ni)
    assert np.array_equal(dec("some", np.float32).reshape(-1, 3), dec("pi", np.float32).reshape(-1, 3)[7:7 + 11 * 100:11])
    assert out["comm"] == {"rank": 0, "world": 1}


@pytest.mark.parametrize("precision", ["fp32", "fp64"])
@pytest.mark.parametrize("world,shape,with_b", [(2, (16, 16, 16), False), (4, (24, 16, 32), True), (8, (16, 16, 64), False)])
def test_slab_decomposition_reproduces_one_gpu_bit_for_bit(fp, eo, precision, world, shape, with_b):
    """`world` ranks, each a handle owning a z-slab (+ ghost planes), stepped as an in-process group
    (the exchange = device-to-device copies; over RCCL it is grouped ncclSend/ncclRecv of the same buffers).  Ghost-plane
    reduce of the int64 charge grid, all-gather of rho, migration every 2 sub-steps.  Against ONE handle holding
    everything: the charge grid, the fields and every particle (matched by its global index) are bit-identical,
    the particle count is conserved, particles did migrate and none outran the ghost planes."""
    rng = np.random.default_rng(world)
    n = 30000
    L = (0.016, 0.016, 0.001 * shape[2])
    dens = 1e15
    spec = box_spec(shape, L, count=n, dt=5e-12, macro_weight=dens * np.prod(L) / n)
    pos = rng.random((n, 3)) * L
    vel = rng.normal(0, 0.02, (n, 3))            # ~0.15 cell per sub-step along z: steady traffic across the slab faces
    ion_n = 5000
    pion, vion = rng.random((ion_n, 3)) * L, rng.normal(0, 1e-3, (ion_n, 3))

    one = fp.makeCylindricalParticlePusher(spec, precision=precision)
    one.addSpecies(MP, -QE, ion_n)
    one.set(position=pos, velocity=vel); one.set(position=pion, velocity=vion, species=1)

    nzl = shape[2] // world
    ranks = []
    for r in range(world):
        s = fp.makeCylindricalParticlePusher(dict(spec, count=n), precision=precision)      # capacity: everything could end up here
        s.addSpecies(MP, -QE, ion_n)
        s.domainInit(r, world, ghost_planes=2, migrate_every=2)
        for sp, (p, v) in enumerate(((pos, vel), (pion, vion))):
            own = np.floor(p[:, 2] / L[2] * shape[2]).astype(int) // nzl == r
            ids = np.nonzero(own)[0]
            s._pending = getattr(s, "_pending", {})
            s._pending[sp] = ids
        ranks.append(s)
    # global indices must be contiguous per rank and species: permute the population so that they are
    order = [np.concatenate([r._pending[sp] for r in ranks]) for sp in (0, 1)]
    pos, vel, pion, vion = pos[order[0]], vel[order[0]], pion[order[1]], vion[order[1]]
    one.set(position=pos, velocity=vel); one.set(position=pion, velocity=vion, species=1)
    for sp, (p, v) in enumerate(((pos, vel), (pion, vion))):
        first = 0
        for r in ranks:
            m = len(r._pending[sp])
            r.domainSet(p[first:first + m], v[first:first + m], first_id=first, species=sp)
            first += m
    if with_b:
        one.addB(0.02, 0.0, 0.05)
        for r in ranks:
            r.addB(0.02, 0.0, 0.05)
    group = fp.BoxGroup(ranks)
    one.precalc(); group.precalc()

    def compare(tag):
        f1 = one.readField(fp.F3_RHO_FIXED).reshape(shape[2], -1)
        for r, s in enumerate(ranks):       # every rank's OWN planes hold the complete charge
            fr = s.readField(fp.F3_RHO_FIXED).reshape(shape[2], -1)
            assert np.array_equal(fr[r * nzl:(r + 1) * nzl], f1[r * nzl:(r + 1) * nzl]), (tag, r)
            assert same_bits(s.readField(fp.F3_RHO), one.readField(fp.F3_RHO)), (tag, r)
            assert same_bits(s.readField(fp.F3_E), one.readField(fp.F3_E)), (tag, r)
        for sp, total in ((0, n), (1, ion_n)):
            parts = [s.domainGet(species=sp) for s in ranks]
            ids = np.concatenate([p["ids"] for p in parts])
            assert len(ids) == total and np.array_equal(np.sort(ids), np.arange(total)), (tag, sp)
            got_p = np.concatenate([p["position"] for p in parts])[np.argsort(ids)]
            got_v = np.concatenate([p["velocity"] for p in parts])[np.argsort(ids)]
            ref = one.getParticles(species=sp)
            assert same_bits(got_p, ref["position"]) and same_bits(got_v, ref["velocity"]), (tag, sp)

    compare("precalc")
    for frame in range(4):
        one.step(); group.step()
        compare("frame %d" % frame)
    stats = [s.domainStats() for s in ranks]
    assert sum(s["migrated"] for s in stats) > 0 and all(s["lost"] == 0 for s in stats)
    with pytest.raises(fp.FusionPicError):
        ranks[0].step()                        # a rank of a multi-rank decomposition cannot step alone
    with pytest.raises(fp.FusionPicError):
        ranks[0].getParticles()
    for s in ranks + [one]:
        s.destroy()


@pytest.mark.parametrize("precision", ["fp32", "fp64"])
@pytest.mark.parametrize("world,shape", [(2, (16, 16, 16)), (4, (32, 16, 32)), (8, (16, 32, 64)), (4, (32, 16, 64)), (8, (64, 32, 128)), (8, (8, 16, 32)),
                                         (2, (32, 32, 256)), (8, (128, 64, 512))])
def test_interface_solve_along_the_decomposed_direction(fp, eo, precision, world, shape):
    """distributed_solve = 2 (csrc/fes_tri.hpp): no transposition — after the x and y transforms of its own planes every
    rank reduces each (kx, ky) mode's periodic tridiagonal system along z to a two-equation interface, ONE all-gather
    carries two planes per rank, and every rank solves the interface system and substitutes back.  On every rank's own
    planes the potential agrees with ONE handle's three-dimensional transform within the solve's tolerance (2e-5 / 1e-10 of
    the largest potential), the field on the slab and its ghost planes likewise; after four frames every particle agrees
    to 1e-4 / 1e-9 of the box; none is lost.  4 to 128 planes per rank, and the longest column the library's transforms
    support (512 planes).  (World sizes that are no power of two: the host test of the core, tests/test_tri_core.py.)"""
    if shape[2] >= 256 and precision == "fp64" and world == 8:
        pytest.skip("the 512-plane case runs once, in fp32")
    rng = np.random.default_rng(300 + world)
    n = 40000
    L = (1e-3 * shape[0], 1.3e-3 * shape[1], 0.8e-3 * shape[2])         # cells that are no cubes: lam carries dz^2 / dx^2
    spec = box_spec(shape, L, count=n, dt=5e-12, macro_weight=1e15 * np.prod(L) / n)
    pos, vel = rng.random((n, 3)) * L, rng.normal(0, 0.02, (n, 3))
    nzl = shape[2] // world
    own = np.floor(pos[:, 2] / L[2] * shape[2]).astype(int) // nzl
    order = np.argsort(own, kind="stable")
    pos, vel, own = pos[order], vel[order], own[order]
    one = fp.makeCylindricalParticlePusher(spec, precision=precision)
    one.set(position=pos, velocity=vel)
    ranks, first = [], 0
    for r in range(world):
        s = fp.makeCylindricalParticlePusher(spec, precision=precision)
        s.domainInit(r, world, ghost_planes=min(2, nzl - 2), migrate_every=2, distributed_solve="interface")
        m = int((own == r).sum())
        s.domainSet(pos[first:first + m], vel[first:first + m], first_id=first)
        first += m
        ranks.append(s)
    group = fp.BoxGroup(ranks)
    one.precalc(); group.precalc()
    tol = 2e-5 if precision == "fp32" else 1e-10
    plane = shape[0] * shape[1]
    G = min(2, nzl - 2)

    def fields(tag, exact_rho):
        f1 = one.readField(fp.F3_RHO_FIXED).reshape(shape[2], plane)
        p1 = one.readField(fp.F3_PHI, np.float64).reshape(shape[2], plane)
        e1 = one.readField(fp.F3_E, np.float64).reshape(shape[2], plane, 4)
        for r, s in enumerate(ranks):
            planes = np.arange(r * nzl, (r + 1) * nzl)
            if exact_rho:
                assert np.array_equal(s.readField(fp.F3_RHO_FIXED).reshape(shape[2], plane)[planes], f1[planes]), (tag, r)
            pr = s.readField(fp.F3_PHI, np.float64).reshape(shape[2], plane)
            assert np.abs(pr[planes] - p1[planes]).max() <= tol * np.abs(p1).max(), (tag, r, np.abs(pr[planes] - p1[planes]).max() / np.abs(p1).max())
            wide = np.arange(r * nzl - G, (r + 1) * nzl + G + 1) % shape[2]           # the slab and its ghost planes
            er = s.readField(fp.F3_E, np.float64).reshape(shape[2], plane, 4)
            assert np.abs(er[wide][..., :3] - e1[wide][..., :3]).max() <= 20 * tol * np.abs(e1[..., :3]).max(), (tag, r)

    fields("precalc", True)
    for frame in range(4):
        one.step(); group.step()
    fields("after 4 frames", False)
    parts = [s.domainGet(np.float64) for s in ranks]
    ids = np.concatenate([p["ids"] for p in parts])
    assert np.array_equal(np.sort(ids), np.arange(n))
    got = np.concatenate([p["position"] for p in parts])[np.argsort(ids)]
    d = np.abs(got - one.getParticles(np.float64)["position"]); d = np.minimum(d, 1 - d)
    assert d.max() <= (1e-4 if precision == "fp32" else 1e-9)
    stats = [s.domainStats() for s in ranks]
    assert all(s["lost"] == 0 for s in stats)
    for s in ranks + [one]:
        s.destroy()


def test_interface_solve_is_refused_where_it_cannot_run(fp):
    """a grid that is no power of two (rocFFT path) and a world of more than eight ranks"""
    s = fp.makeCylindricalParticlePusher(box_spec((24, 16, 48), (1.0, 1.0, 1.0), count=10))
    with pytest.raises(fp.FusionPicError):
        s.domainInit(0, 4, ghost_planes=2, migrate_every=2, distributed_solve="interface")
    s.destroy()
    s = fp.makeCylindricalParticlePusher(box_spec((16, 16, 64), (1.0, 1.0, 1.0), count=10))
    with pytest.raises(fp.FusionPicError):
        s.domainInit(0, 16, ghost_planes=1, migrate_every=2, distributed_solve="interface")
    s.destroy()


@pytest.mark.parametrize("precision", ["fp32", "fp64"])
@pytest.mark.parametrize("world,shape", [(2, (16, 16, 16)), (4, (24, 16, 32)), (8, (20, 32, 64)), (4, (32, 16, 64)), (8, (64, 32, 128))])
def test_slab_decomposed_poisson_solve(fp, eo, precision, world, shape):
    """distributed_solve: no rank transforms the whole grid — 2-D transforms of the owned planes, all-to-all transposition,
    transforms along z on ny/N rows of ky, back, potential ghost planes, gradient on the slab and its ghost planes.  The
    charge grid stays exact; potential and field agree with ONE GPU's 3-D transform to rounding on every rank's own planes
    and ghost planes; after four frames every particle (matched by global index) agrees to the solve's tolerance and none
    is lost."""
    rng = np.random.default_rng(100 + world)
    n = 40000
    L = (1e-3 * shape[0], 1e-3 * shape[1], 1e-3 * shape[2])
    spec = box_spec(shape, L, count=n, dt=5e-12, macro_weight=1e15 * np.prod(L) / n)
    pos, vel = rng.random((n, 3)) * L, rng.normal(0, 0.02, (n, 3))
    nzl = shape[2] // world
    own = np.floor(pos[:, 2] / L[2] * shape[2]).astype(int) // nzl
    order = np.argsort(own, kind="stable")
    pos, vel, own = pos[order], vel[order], own[order]
    one = fp.makeCylindricalParticlePusher(spec, precision=precision)
    one.set(position=pos, velocity=vel)
    ranks, first = [], 0
    for r in range(world):
        s = fp.makeCylindricalParticlePusher(spec, precision=precision)
        s.domainInit(r, world, ghost_planes=2, migrate_every=2, distributed_solve=True)
        m = int((own == r).sum())
        s.domainSet(pos[first:first + m], vel[first:first + m], first_id=first)
        first += m
        ranks.append(s)
    group = fp.BoxGroup(ranks)
    one.precalc(); group.precalc()
    tol = 3e-5 if precision == "fp32" else 1e-10
    plane = shape[0] * shape[1]

    def fields(tag, exact_rho):
        f1 = one.readField(fp.F3_RHO_FIXED).reshape(shape[2], plane)
        p1 = one.readField(fp.F3_PHI, np.float64).reshape(shape[2], plane)
        e1 = one.readField(fp.F3_E, np.float64).reshape(shape[2], plane, 4)
        for r, s in enumerate(ranks):
            planes = np.arange(r * nzl, (r + 1) * nzl)
            if exact_rho:
                assert np.array_equal(s.readField(fp.F3_RHO_FIXED).reshape(shape[2], plane)[planes], f1[planes]), (tag, r)
            pr = s.readField(fp.F3_PHI, np.float64).reshape(shape[2], plane)
            assert np.abs(pr[planes] - p1[planes]).max() <= tol * np.abs(p1).max(), (tag, r)
            wide = np.arange(r * nzl - 2, (r + 1) * nzl + 3) % shape[2]           # the slab and its ghost planes
            er = s.readField(fp.F3_E, np.float64).reshape(shape[2], plane, 4)
            assert np.abs(er[wide][..., :3] - e1[wide][..., :3]).max() <= 20 * tol * np.abs(e1[..., :3]).max(), (tag, r)

    fields("precalc", True)
    for frame in range(4):
        one.step(); group.step()
    fields("after 4 frames", False)
    parts = [s.domainGet(np.float64) for s in ranks]
    ids = np.concatenate([p["ids"] for p in parts])
    assert np.array_equal(np.sort(ids), np.arange(n))
    got = np.concatenate([p["position"] for p in parts])[np.argsort(ids)]
    d = np.abs(got - one.getParticles(np.float64)["position"]); d = np.minimum(d, 1 - d)
    assert d.max() <= (1e-4 if precision == "fp32" else 1e-9)
    stats = [s.domainStats() for s in ranks]
    assert sum(s["migrated"] for s in stats) > 0 and all(s["lost"] == 0 for s in stats)
    for s in ranks + [one]:
        s.destroy()


def _decomposed_run(fp, spec, precision, world, G, pos, vel, frames, species_b=None):
    """a group of `world` in-process ranks with the decomposed solve; returns per-rank read-backs and the particles"""
    shape = (spec["nr"], spec["ny"], spec["nz"])
    L = (spec["radius"], spec["length_y"], spec["height"])
    nzl = shape[2] // world
    own = np.floor(pos[:, 2] / L[2] * shape[2]).astype(int) // nzl
    order = np.argsort(own, kind="stable")
    pos, vel, counts = pos[order], vel[order], np.bincount(own, minlength=world)
    ranks = []
    for r in range(world):
        s = fp.makeCylindricalParticlePusher(dict(spec, count=2 * len(pos)), precision=precision)
        s.domainInit(r, world, ghost_planes=G, migrate_every=2, distributed_solve=True)
        first = int(counts[:r].sum())
        s.domainSet(pos[first:first + counts[r]], vel[first:first + counts[r]], first_id=first)
        ranks.append(s)
    group = fp.BoxGroup(ranks)
    group.precalc()
    for _ in range(frames):
        group.step()
    out = {"grid_bytes": [s.stats()["bytes_grid_state"] for s in ranks], "ranks": []}
    for r, s in enumerate(ranks):
        held = np.arange(r * nzl - G, (r + 1) * nzl + G + 1) % shape[2]     # what the cycle of this rank reads and writes
        grids = {}
        for name, which, kind in (("rho_fixed", fp.F3_RHO_FIXED, None), ("phi", fp.F3_PHI, np.float64), ("E", fp.F3_E, np.float64), ("rho", fp.F3_RHO, np.float64)):
            a = s.readField(which) if kind is None else s.readField(which, kind)
            grids[name] = a.reshape(shape[2], -1)[held if name != "rho" else held[G:G + nzl]]
        part = s.domainGet(np.float64)
        out["ranks"].append((grids, part, s.domainStats()))
    for s in ranks:
        s.destroy()
    return out


@pytest.mark.parametrize("precision", ["fp32", "fp64"])
@pytest.mark.parametrize("world,shape,G", [(4, (16, 32, 64), 2), (8, (32, 16, 128), 3), (4, (16, 16, 32), 1), (2, (8, 16, 64), 2)])
def test_slab_only_arrays_change_nothing(fp, monkeypatch, precision, world, shape, G):
    """A rank of a decomposition with the decomposed solve keeps nzl + 2 (G + 2) + 1 planes of every node array instead of
    the whole grid (fes_api.hip, keep_slab_only).  Both layouts run the same kernels on the same numbers: every grid the
    cycle touches, every particle, the migration counters are bit-identical to the run that keeps whole-grid arrays
    (FPIC_DOMAIN_COMPACT=0), the first and last slab's halo wraps around the box, and the grid memory of a rank shrinks."""
    rng = np.random.default_rng(77 + world)
    n = 30000
    L = tuple(1e-3 * s for s in shape)
    spec = box_spec(shape, L, count=n, dt=5e-12, macro_weight=1e15 * np.prod(L) / n)
    pos, vel = rng.random((n, 3)) * L, rng.normal(0, 0.02, (n, 3))
    monkeypatch.setenv("FPIC_DOMAIN_COMPACT", "0")
    whole = _decomposed_run(fp, spec, precision, world, G, pos, vel, frames=5)
    monkeypatch.delenv("FPIC_DOMAIN_COMPACT")
    slab = _decomposed_run(fp, spec, precision, world, G, pos, vel, frames=5)
    for r in range(world):
        (g0, p0, st0), (g1, p1, st1) = whole["ranks"][r], slab["ranks"][r]
        for name in g0:
            assert same_bits(g0[name], g1[name]) if g0[name].dtype.kind == "f" else np.array_equal(g0[name], g1[name]), (r, name)
        i0, i1 = np.argsort(p0["ids"]), np.argsort(p1["ids"])          # (a rank holds its particles in no particular order)
        assert np.array_equal(p0["ids"][i0], p1["ids"][i1]), r
        assert same_bits(p0["position"][i0], p1["position"][i1]) and same_bits(p0["velocity"][i0], p1["velocity"][i1]), r
        assert st0 == st1, r
    assert sum(st["migrated"] for _, _, st in slab["ranks"]) > 0
    nzs = shape[2] // world + 2 * (G + 2) + 1
    assert nzs < shape[2]
    esz = 4 if precision == "fp32" else 8
    tile = 128 // (2 * esz)                      # the transform buffer's rows are padded to whole column tiles of 128 bytes
    saved = (shape[2] - nzs) * shape[0] * shape[1] * (8 + 6 * esz) + (shape[0] // 2 + tile) // tile * tile * shape[1] * shape[2] * 2 * esz
    assert all(w - c == saved for w, c in zip(whole["grid_bytes"], slab["grid_bytes"]))


@pytest.mark.parametrize("precision", ["fp32", "fp64"])
@pytest.mark.parametrize("shape,counts,with_b", [((32, 32, 16), (60000, 60000), False), ((48, 32, 24), (90001, 2500, 40000), True),
                                                 ((16, 16, 8), (70000, 9), False)])
def test_species_sharing_a_launch(fp, monkeypatch, precision, shape, counts, with_b):
    """Binned species are pushed by ONE launch (a tile's window staged and flushed once for all of them, Push3Joint): over
    re-binnings, with populations of very different sizes (tiles that are empty in one species, tiles of several work items
    in another), every particle and the integer charge grid are those of one launch per species (FPIC_PUSH_JOINT=0), whose
    parity with the oracle the tests above hold."""
    rng = np.random.default_rng(len(counts) * 100 + shape[0])
    L = tuple(1e-3 * s for s in shape)
    spec = box_spec(shape, L, count=counts[0], dt=2e-12, macro_weight=1e12 * np.prod(L) / counts[0])
    masses = [ME, MP, 4 * MP][:len(counts)]
    charges = [QE, -QE, -2 * QE][:len(counts)]
    state = [(rng.random((n, 3)) * L, rng.normal(0, 0.02, (n, 3))) for n in counts]
    state[1][0][:, 2] = rng.random(len(state[1][0])) * L[2] * 0.3          # the second species fills a third of the box: empty tiles

    def run(joint):
        if joint:
            monkeypatch.delenv("FPIC_PUSH_JOINT", raising=False)
        else:
            monkeypatch.setenv("FPIC_PUSH_JOINT", "0")
        sim = fp.makeCylindricalParticlePusher(dict(spec, sort_interval=3), precision=precision)
        for sp in range(1, len(counts)):
            sim.addSpecies(masses[sp], charges[sp], counts[sp])
        for sp, (p, v) in enumerate(state):
            sim.set(position=p, velocity=v, species=sp)
        if with_b:
            sim.addB(0.0, 0.02, 0.05)
        sim.precalc()
        out = []
        for _ in range(4):
            sim.step()
            out.append(([sim.getParticles(species=sp) for sp in range(len(counts))], sim.readField(fp.F3_RHO_FIXED)))
        sim.destroy()
        return out

    together, apart = run(True), run(False)
    for frame, ((pa, ra), (pb, rb)) in enumerate(zip(together, apart)):
        assert np.array_equal(ra, rb), frame
        for sp in range(len(counts)):
            assert same_bits(pa[sp]["position"], pb[sp]["position"]) and same_bits(pa[sp]["velocity"], pb[sp]["velocity"]), (frame, sp)


# ---------------------------------------------------------------------------- LDS-staged (two-level) first binning
# Populations of 2^20 particles and more are binned by sort_scatter_kernel; FPIC_TWO_LEVEL_MIN (read when the handle is
# created) lowers that size so that the small oracle-checked scenes above run through the same kernels: ragged last
# chunks, empty coarse bins, dead slots of a decomposed rank, fp64 stage, the 8^3 tiles of the Yee mode.

@pytest.mark.parametrize("precision", ["fp32", "fp64"])
@pytest.mark.parametrize("shape,n", [((16, 16, 8), 4000), ((72, 40, 48), 30011), ((160, 16, 16), 5000)])
def test_staged_binning_on_small_scenes(fp, eo, monkeypatch, precision, shape, n):
    monkeypatch.setenv("FPIC_TWO_LEVEL_MIN", "1")
    test_push_and_deposit_bit_exact_in_a_given_field(fp, eo, precision, shape, n, True, 1)


@pytest.mark.parametrize("precision", ["fp32", "fp64"])
def test_staged_binning_of_a_decomposed_box(fp, eo, monkeypatch, precision):
    monkeypatch.setenv("FPIC_TWO_LEVEL_MIN", "1")
    test_slab_decomposition_reproduces_one_gpu_bit_for_bit(fp, eo, precision, 4, (72, 40, 96), True)


def test_two_level_binning_at_a_million_particles_is_bit_exact(fp, eo):
    """2^20 + 777 electrons on a 64 x 64 x 32 grid (65 tiles: coarse groups of 9): the production size threshold, no
    override; push + integer deposit bit-exact against the oracle after a first binning and two re-binnings."""
    test_push_and_deposit_bit_exact_in_a_given_field(fp, eo, "fp32", (64, 64, 32), (1 << 20) + 777, True, 2)


def test_decomposed_rank_through_the_javascript_host(fp, eo, tmp_path):
    """The decomposition entry points through Node: this process is rank 0 of a world of one (the same calls every
    rank of an N-process run makes: commInit, domainInit, domainSet with global indices, precalc, step, domainGet).
    Against a plain handle on the same particles: the charge grid and, matched by index, every particle bit-identical."""
    import base64
    import json
    import os
    import shutil
    import subprocess
    from helpers import ROOT
    node = shutil.which("node")
    if node is None:
        pytest.skip("node is not installed on this box")
    rng = np.random.default_rng(5)
    n = 5000
    shape, L = (16, 16, 16), (0.016, 0.016, 0.016)
    spec = box_spec(shape, L, count=n, dt=5e-12, macro_weight=1e15 * np.prod(L) / n)
    pos, vel = rng.random((n, 3)) * L, rng.normal(0, 0.02, (n, 3))
    (tmp_path / "in.json").write_text(json.dumps(dict(spec=spec, pos=pos.tolist(), vel=vel.tolist())))
    script = r"""
const fs = require('fs');
const empic = require(process.argv[1]);
const inp = JSON.parse(fs.readFileSync(process.argv[2]));
const sim = empic.makeCylindricalParticlePusher(inp.spec);
sim.commInit(empic.commUniqueId(), 0, 1);
sim.domainInit(0, 1, {ghost_planes: 2, migrate_every: 2});
sim.domainSet({position: inp.pos, velocity: inp.vel}, 0);
sim.precalc();
for (let frame = 0; frame < 3; frame++) sim.step();
const got = sim.domainGet();
const b64 = a => Buffer.from(a.buffer, a.byteOffset, a.byteLength).toString('base64');
let err = 'none';
try { sim.domainSet({position: new Float32Array(9), velocity: new Float32Array(6)}, 0); } catch (x) { err = x.constructor.name; }
console.log(JSON.stringify({n: got.n, pos: b64(got.position), vel: b64(got.velocity), ids: b64(got.ids), fixed: b64(sim.readField('rho_fixed')),
  stats: sim.domainStats(), err: err}));
sim.destroy();
"""
    shim = os.path.join(ROOT, "fusion-sim_amd", "js", "empic_native.js")
    raw = subprocess.check_output([node, "-e", script, shim, str(tmp_path / "in.json")])
    out = json.loads(raw.decode().strip().splitlines()[-1])
    dec = lambda k, dt: np.frombuffer(base64.b64decode(out[k]), dtype=dt)
    one = fp.makeCylindricalParticlePusher(spec)
    one.set(position=pos, velocity=vel)
    one.precalc()
    one.step(3)
    ref = one.getParticles()
    ids = dec("ids", np.uint32)
    assert out["n"] == n and np.array_equal(np.sort(ids), np.arange(n))
    order = np.argsort(ids)
    assert same_bits(dec("pos", np.float32).reshape(-1, 3)[order], ref["position"])
    assert same_bits(dec("vel", np.float32).reshape(-1, 3)[order], ref["velocity"])
    assert np.array_equal(dec("fixed", np.int64), one.readField(fp.F3_RHO_FIXED))
    assert out["err"] == "RangeError" and out["stats"]["lost"] == 0
    one.destroy()


@pytest.mark.parametrize("precision", ["fp32", "fp64"])
def test_decomposition_with_an_empty_and_an_emptied_rank(fp, eo, precision):
    """Two species in a 3-rank decomposition: the electrons all start in slab 0 and stream upwards (slab 0 empties, slab 1
    — empty at first — fills, later slab 2), the ions sit in slab 2 only.  Ranks without particles of a species, ranks
    that lose all of them and ranks whose first particles arrive by migration, against one handle: bit-identical."""
    rng = np.random.default_rng(3)
    world, shape = 3, (16, 16, 24)
    L = (0.016, 0.016, 0.024)
    n, ni = 9000, 2000
    dt = 5e-12
    spec = box_spec(shape, L, count=n, dt=dt, solver="none", macro_weight=1e4)
    nzl = shape[2] // world
    cell_per_substep = 0.3                                               # along z, every electron
    vz = cell_per_substep * (L[2] / shape[2]) / (dt * 2.998e8)
    pos = rng.random((n, 3)) * (L[0], L[1], L[2] * 5.0 / shape[2]) + (0, 0, L[2] * 1.0 / shape[2])   # planes 1..6 of slab 0
    vel = np.concatenate([rng.normal(0, 0.01, (n, 2)), np.full((n, 1), vz)], axis=1)
    pi = rng.random((ni, 3)) * (L[0], L[1], L[2] / world) + (0, 0, 2 * L[2] / world)
    vi = rng.normal(0, 1e-4, (ni, 3))
    E = rng.normal(0, 1e3, shape + (3,))
    one = fp.makeCylindricalParticlePusher(spec, precision=precision)
    one.addSpecies(MP, -QE, ni)
    one.set(position=pos, velocity=vel, E=E); one.set(position=pi, velocity=vi, species=1)
    ranks = []
    for r in range(world):
        s = fp.makeCylindricalParticlePusher(spec, precision=precision)
        s.addSpecies(MP, -QE, ni)
        s.domainInit(r, world, ghost_planes=2, migrate_every=2)
        s.set(E=E)
        if r == 0:
            s.domainSet(pos, vel, first_id=0)
        if r == 2:
            s.domainSet(pi, vi, first_id=0, species=1)
        ranks.append(s)
    group = fp.BoxGroup(ranks)
    one.precalc(); group.precalc()
    held = []
    for frame in range(24):                                               # 48 sub-steps x 0.3 cells = 14.4 cells: through slab 1 into slab 2
        one.step(); group.step()
        f1 = one.readField(fp.F3_RHO_FIXED).reshape(shape[2], -1)
        for r, s in enumerate(ranks):
            fr = s.readField(fp.F3_RHO_FIXED).reshape(shape[2], -1)
            assert np.array_equal(fr[r * nzl:(r + 1) * nzl], f1[r * nzl:(r + 1) * nzl]), (frame, r)
        for sp, total in ((0, n), (1, ni)):
            parts = [s.domainGet(species=sp) for s in ranks]
            if sp == 0:
                held.append([len(p["ids"]) for p in parts])
            ids = np.concatenate([p["ids"] for p in parts])
            assert np.array_equal(np.sort(ids), np.arange(total)), (frame, sp)
            ref = one.getParticles(species=sp)
            assert same_bits(np.concatenate([p["position"] for p in parts])[np.argsort(ids)], ref["position"]), (frame, sp)
            assert same_bits(np.concatenate([p["velocity"] for p in parts])[np.argsort(ids)], ref["velocity"]), (frame, sp)
    held = np.array(held)
    assert held[0, 0] > 0 and held[-1, 0] == 0, "slab 0 must have emptied"
    assert held[0, 1] < n // 2 and held[:, 1].max() > n // 2 and held[-1, 2] > 0, "slab 1 fills from nothing, slab 2 receives later"
    assert all(s.domainStats()["lost"] == 0 for s in ranks)
    for s in ranks + [one]:
        s.destroy()


@pytest.mark.parametrize("precision", ["fp32", "fp64"])
@pytest.mark.parametrize("solver", ["poisson_fft", "none", "yee"])
def test_box_checkpoint_resume_is_bit_identical(fp, eo, tmp_path, precision, solver):
    """save after 3 frames, 3 more frames; a fresh handle restored from the file and stepped 3 frames: particles of both
    species, the charge (or current) grid and the fields are bit-identical.  A truncated file, a file of another box and
    a decomposed handle are refused, the handle unchanged."""
    rng = np.random.default_rng(21)
    shape, L = (20, 16, 24), (0.02, 0.016, 0.024)
    n, ni = 30000, 8000
    dt = 5e-12
    if solver == "yee":
        d = [L[a] / shape[a] for a in range(3)]
        dt = 0.5 / (2.998e8 * np.sqrt(sum(1 / x ** 2 for x in d)))
    spec = box_spec(shape, L, count=n, dt=dt, solver=solver, macro_weight=1e15 * np.prod(L) / n)
    pos, vel = rng.random((n, 3)) * L, rng.normal(0, 0.02, (n, 3))
    pi, vi = rng.random((ni, 3)) * L, rng.normal(0, 1e-3, (ni, 3))
    E = rng.normal(0, 3e4, shape + (3,))

    def build():
        s = fp.makeCylindricalParticlePusher(spec, precision=precision, sort_interval=2 if solver != "yee" else 0)
        s.addSpecies(MP, -QE, ni)
        return s

    a = build()
    a.set(position=pos, velocity=vel); a.set(position=pi, velocity=vi, species=1)
    if solver == "none":
        a.set(E=E)
    a.addB(0.01, -0.02, 0.05)
    a.precalc()
    a.step(3)
    path = tmp_path / "box.ckpt"
    a.saveCheckpoint(path)
    a.step(3)

    b = build()
    b.set(position=pos[::-1], velocity=vel[::-1])     # something else, to be overwritten
    b.loadCheckpoint(path)
    b.step(3)
    for sp in (0, 1):
        ga, gb = a.getParticles(species=sp), b.getParticles(species=sp)
        assert same_bits(ga["position"], gb["position"]) and same_bits(ga["velocity"], gb["velocity"]), sp
    fields = [fp.F3_RHO_FIXED, fp.F3_E] + ([fp.F3_EDGE_E, fp.F3_FACE_B, fp.F3_J_FIXED] if solver == "yee" else [])
    if solver == "yee":
        a.density(); b.density()
    for which in fields:
        fa, fb = a.readField(which), b.readField(which)
        assert np.array_equal(fa, fb) if fa.dtype == np.int64 else same_bits(fa, fb), which

    # refusals
    data = path.read_bytes()
    (tmp_path / "short.ckpt").write_bytes(data[:len(data) - 1000])
    before = b.getParticles()
    with pytest.raises(fp.FusionPicError, match="truncated"):
        b.loadCheckpoint(tmp_path / "short.ckpt")
    (tmp_path / "old.ckpt").write_bytes(data[:8] + (0).to_bytes(4, "little") + data[12:])
    with pytest.raises(fp.FusionPicError, match="format version 0"):
        b.loadCheckpoint(tmp_path / "old.ckpt")          # a file of another format: refused by its version, not as "truncated"
    (tmp_path / "v1.ckpt").write_bytes(data[:8] + (1).to_bytes(4, "little") + data[12:])
    c1, c2 = build(), build()
    c1.loadCheckpoint(tmp_path / "v1.ckpt")              # version 1 had the same layout (round 2's files): read
    c2.loadCheckpoint(path)
    assert same_bits(c1.getParticles()["position"], c2.getParticles()["position"])
    c1.destroy(); c2.destroy()
    other = fp.makeCylindricalParticlePusher(dict(spec, nz=shape[2] + 8), precision=precision)
    with pytest.raises(fp.FusionPicError):
        other.loadCheckpoint(path)                       # one species, another grid
    assert same_bits(b.getParticles()["position"], before["position"])
    dec = fp.makeCylindricalParticlePusher(spec, precision=precision)
    dec.domainInit(0, 2, ghost_planes=2, migrate_every=2)
    with pytest.raises(fp.FusionPicError):
        dec.loadCheckpoint(path)                         # a rank of a decomposition reads a rank's file, not a whole box
    for s in (a, b, other, dec):
        s.destroy()


def test_plasma_box_example_from_node():
    """examples/plasma_box_node.js: a cold plasma oscillation in the box, driven from Node; the frequency read off the
    field's zero crossings is the scheme's omega_p cos(k dx / 2) within 1 %."""
    import json
    import os
    import shutil
    import subprocess
    from helpers import ROOT
    node = shutil.which("node")
    if node is None:
        pytest.skip("node is not installed on this box")
    out = subprocess.check_output([node, os.path.join(ROOT, "examples", "plasma_box_node.js"), "--grid", "16", "--perCell", "8", "--frames", "150"])
    res = json.loads(out.decode().strip().splitlines()[-1])
    assert res["particles"] == 16 ** 3 * 8 and res["updates"] == 300 * res["particles"]
    assert res["relative_error"] < 1e-2, res
    out = subprocess.check_output([node, os.path.join(ROOT, "examples", "plasma_box_node.js"), "--grid", "16", "--perCell", "8", "--frames", "400", "--solver", "yee"])
    res = json.loads(out.decode().strip().splitlines()[-1])
    assert res["relative_error"] < 1e-2 and res["nonzero_current_entries"] > 0, res       # full EM, J_fixed through the addon


def test_large_decomposed_box_is_bit_identical_to_one_handle(fp, eo):
    """2e7 electrons + 4e6 ions on 128^3 nodes over 8 in-process ranks (each rank's population is above the 2^20 limit of
    the staged two-level binning, migration rides on the fused re-binning, slab-decomposed solve off so that the fields
    are bit-comparable): after 6 frames the charge grid on every rank's planes, the field and every particle are
    bit-identical to one handle's; particles migrated, none outran the ghost planes."""
    world, shape = 8, (128, 128, 128)
    L = (0.128, 0.128, 0.128)
    n, ni = 20_000_000, 4_000_000
    spec = box_spec(shape, L, count=n, dt=2e-12, macro_weight=1e15 * np.prod(L) / n)
    rng = np.random.Generator(np.random.Philox(77))
    nzl = shape[2] // world

    def population(m, vth):
        # sorted by slab so that global indices are contiguous per rank
        z = np.sort(rng.random(m, dtype=np.float32)) * np.float32(L[2] * (1 - 1e-6))
        p = np.stack([rng.random(m, dtype=np.float32) * np.float32(L[0]), rng.random(m, dtype=np.float32) * np.float32(L[1]), z], axis=1)
        v = rng.standard_normal((m, 3), dtype=np.float32) * np.float32(vth)
        counts = np.bincount(np.floor(z.astype(np.float64) / L[2] * shape[2]).astype(int) // nzl, minlength=world)
        return p, v, counts

    pe, ve, ce = population(n, 0.02)          # ~0.12 cells per sub-step along z
    pi, vi, ci = population(ni, 5e-4)
    one = fp.makeCylindricalParticlePusher(spec)
    one.addSpecies(MP, -QE, ni)
    one.set(position=pe, velocity=ve); one.set(position=pi, velocity=vi, species=1)
    ranks = []
    for r in range(world):
        s = fp.makeCylindricalParticlePusher(dict(spec, count=int(ce.max() * 1.2)))
        s.addSpecies(MP, -QE, int(ci.max() * 1.2))
        s.domainInit(r, world, ghost_planes=2, migrate_every=4)
        fe, fi = int(ce[:r].sum()), int(ci[:r].sum())
        s.domainSet(pe[fe:fe + ce[r]], ve[fe:fe + ce[r]], first_id=fe)
        s.domainSet(pi[fi:fi + ci[r]], vi[fi:fi + ci[r]], first_id=fi, species=1)
        ranks.append(s)
    group = fp.BoxGroup(ranks)
    one.precalc(); group.precalc()
    for _ in range(6):
        one.step(); group.step()
    f1 = one.readField(fp.F3_RHO_FIXED).reshape(shape[2], -1)
    e1 = one.readField(fp.F3_E)
    for r, s in enumerate(ranks):
        fr = s.readField(fp.F3_RHO_FIXED).reshape(shape[2], -1)
        assert np.array_equal(fr[r * nzl:(r + 1) * nzl], f1[r * nzl:(r + 1) * nzl]), r
        assert same_bits(s.readField(fp.F3_E), e1), r
    for sp, total in ((0, n), (1, ni)):
        parts = [s.domainGet(species=sp) for s in ranks]
        ids = np.concatenate([p["ids"] for p in parts])
        order = np.argsort(ids)
        assert len(ids) == total and np.array_equal(ids[order], np.arange(total, dtype=np.uint32)), sp
        ref = one.getParticles(species=sp)
        assert same_bits(np.concatenate([p["position"] for p in parts])[order], ref["position"]), sp
        assert same_bits(np.concatenate([p["velocity"] for p in parts])[order], ref["velocity"]), sp
    stats = [s.domainStats() for s in ranks]
    assert sum(s["migrated"] for s in stats) > 10000 and all(s["lost"] == 0 for s in stats)
    assert int(f1.astype(object).sum()) == (n - ni) * eo.FIXED_ONE
    for s in ranks + [one]:
        s.destroy()


@pytest.mark.parametrize("seed", list(range(16)))
def test_decomposition_randomised(fp, eo, monkeypatch, seed):
    """Random decompositions against one handle, frame by frame: world 2..8, 1..3 ghost planes, migration every 1..5
    sub-steps, one or two species, with or without B, either precision, populations spread unevenly over the slabs (some
    slabs nearly empty), the staged binning forced on half of the cases.  Velocities are bounded so that no particle can
    outrun the ghost planes between two migrations; everything must stay bit-identical."""
    rng = np.random.default_rng(1000 + seed)
    if seed % 2:
        monkeypatch.setenv("FPIC_TWO_LEVEL_MIN", "1")
    precision = "fp32" if rng.random() < 0.6 else "fp64"
    world = int(rng.choice([2, 3, 4, 6, 8]))
    G = int(rng.integers(1, 4))
    every = int(rng.integers(1, 6))
    nzl = int(rng.integers((2 * G + 1) if world == 2 else (G + 1), 12))   # (two slabs: ghost ranges must not overlap)
    shape = (int(rng.integers(6, 40)), int(rng.integers(6, 40)), world * nzl)
    L = tuple(1e-3 * s for s in shape)
    dt = 5e-12
    n = int(rng.integers(2000, 40000))
    spec = box_spec(shape, L, count=n, dt=dt, macro_weight=1e15 * np.prod(L) / n, solver="poisson_fft" if rng.random() < 0.7 else "none")
    # |v_z| dt c * every < G cells  (with a margin); x and y are free
    vmax = 0.8 * G * 1e-3 / (every * dt * 2.998e8)
    two = rng.random() < 0.5
    pops = []
    for m in ((n, n // 3) if two else (n,)):
        # uneven over the slabs: a power of a uniform deviate piles the particles up at low z
        z = (rng.random(m) ** float(rng.choice([1.0, 2.5, 6.0]))) * L[2] * (1 - 1e-9)
        p = np.stack([rng.random(m) * L[0], rng.random(m) * L[1], z], axis=1)
        v = np.stack([rng.normal(0, 0.05, m), rng.normal(0, 0.05, m), rng.uniform(-vmax, vmax, m)], axis=1)
        owner = np.floor(p[:, 2] / L[2] * shape[2]).astype(int) // nzl
        order = np.argsort(owner, kind="stable")
        pops.append((p[order], v[order], np.bincount(owner, minlength=world)))
    E = rng.normal(0, 2e4, shape + (3,))
    with_b = rng.random() < 0.5

    def build(count, r=None):
        s = fp.makeCylindricalParticlePusher(dict(spec, count=count), precision=precision)
        if two:
            s.addSpecies(MP, -QE, n // 3)
        if r is not None:
            s.domainInit(r, world, ghost_planes=G, migrate_every=every, distributed_solve=False)
        if spec["solver"] == "none":
            s.set(E=E)
        if with_b:
            s.addB(0.02, -0.01, 0.05)
        return s

    one = build(n)
    for sp, (p, v, _) in enumerate(pops):
        one.set(position=p, velocity=v, species=sp)
    ranks = []
    for r in range(world):
        s = build(4 * n, r)      # (capacity; the migration messages hold a quarter of it)
        for sp, (p, v, c) in enumerate(pops):
            first = int(c[:r].sum())
            if c[r]:
                s.domainSet(p[first:first + c[r]], v[first:first + c[r]], first_id=first, species=sp)
        ranks.append(s)
    group = fp.BoxGroup(ranks)
    one.precalc(); group.precalc()
    for frame in range(5):
        one.step(); group.step()
        f1 = one.readField(fp.F3_RHO_FIXED).reshape(shape[2], -1)
        for r, s in enumerate(ranks):
            fr = s.readField(fp.F3_RHO_FIXED).reshape(shape[2], -1)
            assert np.array_equal(fr[r * nzl:(r + 1) * nzl], f1[r * nzl:(r + 1) * nzl]), (frame, r)
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


def test_two_slabs_refuse_overlapping_ghost_ranges(fp):
    s = fp.makeCylindricalParticlePusher(box_spec((8, 8, 8), (1.0, 1.0, 1.0), count=10))
    with pytest.raises(fp.FusionPicError, match="at most 1 ghost planes"):
        s.domainInit(0, 2, ghost_planes=2, migrate_every=1)          # slabs of 4 planes: 2 below + 3 above do not fit
    s.domainInit(0, 2, ghost_planes=1, migrate_every=1)
    s.destroy()


@pytest.mark.parametrize("shape,n", [((8, 8, 8), 1), ((8, 8, 8), 4097), ((16, 16, 8), 4096), ((200, 8, 8), 50000), ((8, 8, 400), 50001),
                                     ((130, 70, 50), 123457), ((256, 256, 16), 300000), ((48, 48, 48), 8191)])
def test_staged_binning_keeps_every_particle(fp, monkeypatch, shape, n):
    """sort() through the staged one- / two-level scatter (forced): read-back in the caller's order is unchanged bit for
    bit, and the cells after the binning are the cells before — for one tile, one chunk exactly, long thin grids (many
    coarse groups nearly empty), populations of one particle and of chunk size +- 1."""
    monkeypatch.setenv("FPIC_TWO_LEVEL_MIN", "1")
    rng = np.random.default_rng(n)
    L = tuple(1e-3 * s for s in shape)
    sim = fp.makeCylindricalParticlePusher(box_spec(shape, L, count=n, solver="none"))
    pos = rng.random((n, 3)) * L
    if n > 10:
        pos[: n // 3] = pos[0] + rng.random((n // 3, 3)) * 1e-4                 # a third of them crowd one tile
    vel = rng.normal(0, 0.01, (n, 3))
    sim.set(position=pos, velocity=vel)
    before, cells = sim.getParticles(), sim.getCells()
    sim.sort()
    after = sim.getParticles()
    assert same_bits(before["position"], after["position"]) and same_bits(before["velocity"], after["velocity"])
    assert np.array_equal(cells, sim.getCells())
    sim.precalc(); sim.step()
    assert int(sim.readField(fp.F3_RHO_FIXED).astype(object).sum()) == n * 2 ** 42
    sim.sort()
    assert sim.getParticles()["position"].shape == (n, 3)
    sim.destroy()


@pytest.mark.parametrize("seed", list(range(8)))
def test_distributed_solve_randomised(fp, eo, seed):
    """Random slab-decomposed solves: world 2..8, grids whose rows divide over the ranks, odd and even nx, 1..3 ghost
    planes, either precision.  Against one handle: the charge grid exact; phi and E within the two transforms' rounding;
    particles after three frames within 1e-4 / 1e-9 of a cell."""
    rng = np.random.default_rng(3000 + seed)
    precision = "fp32" if rng.random() < 0.5 else "fp64"
    world = int(rng.choice([2, 3, 4, 8]))
    G = int(rng.integers(1, 4))
    nzl = int(rng.integers(max(G + 2, 2 * G + 1 if world == 2 else 0), 11))   # (the decomposed solve needs G + 2 planes per slab)
    shape = (int(rng.integers(5, 30)), world * int(rng.integers(1, 6)), world * nzl)
    L = tuple(1e-3 * s for s in shape)
    n = int(rng.integers(3000, 30000))
    spec = box_spec(shape, L, count=n, dt=5e-12, macro_weight=1e15 * np.prod(L) / n)
    pos = rng.random((n, 3)) * L
    vel = rng.normal(0, 0.01, (n, 3))
    owner = np.floor(pos[:, 2] / L[2] * shape[2]).astype(int) // nzl
    order = np.argsort(owner, kind="stable")
    pos, vel, counts = pos[order], vel[order], np.bincount(owner, minlength=world)
    one = fp.makeCylindricalParticlePusher(spec, precision=precision)
    one.set(position=pos, velocity=vel)
    ranks = []
    for r in range(world):
        s = fp.makeCylindricalParticlePusher(dict(spec, count=2 * n), precision=precision)
        s.domainInit(r, world, ghost_planes=G, migrate_every=2, distributed_solve=True)
        first = int(counts[:r].sum())
        s.domainSet(pos[first:first + counts[r]], vel[first:first + counts[r]], first_id=first)
        ranks.append(s)
    group = fp.BoxGroup(ranks)
    one.precalc(); group.precalc()
    tol = 5e-5 if precision == "fp32" else 1e-10
    e1 = one.readField(fp.F3_E, np.float64).reshape(shape[2], -1, 4)
    scale = np.abs(e1[..., :3]).max()
    for r, s in enumerate(ranks):
        assert np.array_equal(s.readField(fp.F3_RHO_FIXED).reshape(shape[2], -1)[r * nzl:(r + 1) * nzl],
                              one.readField(fp.F3_RHO_FIXED).reshape(shape[2], -1)[r * nzl:(r + 1) * nzl])
        er = s.readField(fp.F3_E, np.float64).reshape(shape[2], -1, 4)[r * nzl:(r + 1) * nzl]
        assert np.abs(er[..., :3] - e1[r * nzl:(r + 1) * nzl, :, :3]).max() <= tol * scale, r
    for _ in range(3):
        one.step(); group.step()
    parts = [s.domainGet(np.float64) for s in ranks]
    ids = np.concatenate([p["ids"] for p in parts])
    idx = np.argsort(ids)
    assert np.array_equal(ids[idx], np.arange(n))
    d = np.abs(np.concatenate([p["position"] for p in parts])[idx] - one.getParticles(np.float64)["position"])
    d = np.minimum(d, 1 - d)
    assert d.max() <= (1e-4 if precision == "fp32" else 1e-9)
    for s in ranks + [one]:
        s.destroy()


@pytest.mark.parametrize("seed", list(range(14)))
def test_box_randomised_against_the_oracle(fp, eo, monkeypatch, seed):
    """Random boxes in a given field (solver 'none': everything integer or bit-comparable): grids from 2 x 2 x 2 to sizes
    that are no multiple of the 16 x 16 x 8 tile, 1..3 species of random charge number and mass, with or without B,
    sort_interval 0..3, either precision, fast particles (several cells per sub-step for some), the staged binning forced
    on every third case; particles, cells and the int64 charge grid bit-identical to the oracle's after every frame."""
    rng = np.random.default_rng(4000 + seed)
    if seed % 3 == 0:
        monkeypatch.setenv("FPIC_TWO_LEVEL_MIN", "1")
    precision = "fp32" if rng.random() < 0.5 else "fp64"
    dtype = np.float32 if precision == "fp32" else np.float64
    shape = tuple(int(x) for x in rng.integers(2, [70, 50, 40]))
    L = tuple(float(x) for x in rng.uniform(0.5e-3, 2e-3, 3) * shape)
    n = int(rng.integers(1, 30000))
    spec = box_spec(shape, L, count=n, dt=float(rng.uniform(1e-12, 2e-11)), solver="none", macro_weight=float(rng.uniform(1, 1e5)))
    sort_interval = int(rng.integers(0, 4))
    sim = fp.makeCylindricalParticlePusher(spec, precision=precision, sort_interval=sort_interval)
    ora = eo.OracleES3D(spec, dtype)
    counts = [n]
    for _ in range(int(rng.integers(0, 3))):
        m, z, mass = int(rng.integers(1, 8000)), int(rng.choice([-2, -1, 1, 2, 3])), float(rng.choice([ME, MP, 4 * MP]))
        assert sim.addSpecies(mass, z * QE, m) == ora.add_species(mass, z * QE, m)
        counts.append(m)
    for sp, m in enumerate(counts):
        p = rng.random((m, 3)) * L
        v = rng.normal(0, 0.02, (m, 3)) + rng.normal(0, 0.4, (m, 3)) * (rng.random((m, 1)) < 0.2)
        sim.set(position=p, velocity=v, species=sp); ora.set(position=p, velocity=v, species=sp)
    E = rng.normal(0, 5e4, shape + (3,))
    sim.set(E=E); ora.set(E=E)
    if rng.random() < 0.6:
        b = rng.normal(0, 0.3, 3)
        sim.addB(*b); ora.add_b(*b)
    sim.precalc(); ora.precalc()
    assert np.array_equal(sim.readField(fp.F3_RHO_FIXED), ora.rho_fixed)
    for frame in range(4):
        sim.step(); ora.step()
        for sp in range(len(counts)):
            assert_same_particles(sim, ora, species=sp, what="seed %d frame %d species %d" % (seed, frame, sp))
        assert np.array_equal(sim.readField(fp.F3_RHO_FIXED), ora.rho_fixed), (seed, frame)
    sim.destroy()


@pytest.mark.parametrize("precision", ["fp32", "fp64"])
@pytest.mark.parametrize("solver", ["poisson_fft", "yee"])
def test_decomposed_checkpoint_resume_is_bit_identical(fp, eo, tmp_path, precision, solver):
    """Every rank of a 3-rank decomposition writes its own file after 3 frames; the run goes on for 4 frames.  A fresh
    group restored from the files (electrostatic: precalc() recomputes the field from the particles; full EM: the own
    planes come from the file, the halos from the neighbours) and stepped 4 frames holds the same particles — by global
    index, bit for bit — and the same fields on every rank's planes.  A rank refuses another rank's file."""
    rng = np.random.default_rng(8)
    world, shape = 3, (14, 12, 30)
    L = tuple(1e-3 * s for s in shape)
    n, ni = 20000, 5000
    em = solver == "yee"
    dt = 5e-12
    if em:
        d = [L[a] / shape[a] for a in range(3)]
        dt = 0.5 / (2.998e8 * np.sqrt(sum(1 / x ** 2 for x in d)))
    spec = box_spec(shape, L, count=2 * n, dt=dt, solver=solver, macro_weight=1e15 * np.prod(L) / n)
    nzl = shape[2] // world
    pops = []
    for m, vth in ((n, 0.03), (ni, 1e-3)):
        p, v = rng.random((m, 3)) * L, rng.normal(0, vth, (m, 3))
        owner = np.floor(p[:, 2] / L[2] * shape[2]).astype(int) // nzl
        order = np.argsort(owner, kind="stable")
        pops.append((p[order], v[order], np.bincount(owner, minlength=world)))

    def build(fill):
        ranks = []
        for r in range(world):
            s = fp.makeCylindricalParticlePusher(spec, precision=precision)
            s.addSpecies(MP, -QE, 2 * ni)
            s.domainInit(r, world, ghost_planes=2, migrate_every=2)
            if fill:
                for sp, (p, v, c) in enumerate(pops):
                    first = int(c[:r].sum())
                    s.domainSet(p[first:first + c[r]], v[first:first + c[r]], first_id=first, species=sp)
            s.addB(0.0, 0.01, 0.03)
            ranks.append(s)
        return ranks, fp.BoxGroup(ranks)

    a, ga = build(True)
    ga.precalc()
    ga.step(3)
    for r, s in enumerate(a):
        s.saveCheckpoint(tmp_path / ("rank%d.ckpt" % r))
    ga.step(4)

    b, gb = build(False)
    with pytest.raises(fp.FusionPicError, match="rank 1 of 3"):
        b[0].loadCheckpoint(tmp_path / "rank1.ckpt")
    for r, s in enumerate(b):
        s.loadCheckpoint(tmp_path / ("rank%d.ckpt" % r))
    if not em:
        with pytest.raises(fp.FusionPicError, match="precalc"):
            gb.step()
        gb.precalc()
    gb.step(4)
    fields = [fp.F3_EDGE_E, fp.F3_FACE_B] if em else [fp.F3_RHO_FIXED, fp.F3_E]
    for r in range(world):
        for sp in (0, 1):
            pa, pb = a[r].domainGet(species=sp), b[r].domainGet(species=sp)
            ia, ib = np.argsort(pa["ids"]), np.argsort(pb["ids"])
            assert np.array_equal(pa["ids"][ia], pb["ids"][ib]), (r, sp)
            assert same_bits(pa["position"][ia], pb["position"][ib]) and same_bits(pa["velocity"][ia], pb["velocity"][ib]), (r, sp)
        for which in fields:
            fa = a[r].readField(which).reshape(shape[2], -1)[r * nzl:(r + 1) * nzl]
            fb = b[r].readField(which).reshape(shape[2], -1)[r * nzl:(r + 1) * nzl]
            assert np.array_equal(fa, fb) if fa.dtype == np.int64 else same_bits(fa, fb), (r, which)
    data = (tmp_path / "rank0.ckpt").read_bytes()
    (tmp_path / "short.ckpt").write_bytes(data[:-100])
    with pytest.raises(fp.FusionPicError, match="truncated"):
        b[0].loadCheckpoint(tmp_path / "short.ckpt")
    for s in a + b:
        s.destroy()


@pytest.mark.parametrize("solver", ["poisson_fft", "yee"])
def test_a_refused_migration_moves_nothing(fp, solver):
    """ADVICE r03 (low), closed in round 4: the ranks agree ONCE, over every species, whether a migration can be held — before
    the first particle of any species is touched.  Two ranks, two species: the electrons spread evenly and migrate happily,
    the ions all stream upwards out of rank 0 into rank 1, whose ion capacity is a few particles above what it starts with.
    The step whose migration would overfill rank 1 raises on the whole group, with the capacity in the message — and every
    rank still holds exactly the particles (both species, bit for bit) it held before that call: rounds 2-3 agreed species
    by species, so the electrons had already moved when the ions were refused."""
    rng = np.random.default_rng(17)
    world, shape = 2, (16, 16, 32)
    L = tuple(1e-3 * s for s in shape)
    n, ni = 6000, 4000
    nzl = shape[2] // world
    em = solver == "yee"
    dt = 5e-12
    if em:
        d = [L[a] / shape[a] for a in range(3)]
        dt = 0.5 / (2.998e8 * np.sqrt(sum(1 / x ** 2 for x in d)))
    spec = box_spec(shape, L, count=2 * n, dt=dt, solver=solver, macro_weight=1.0)
    pe, ve = rng.random((n, 3)) * L, rng.normal(0, 0.01, (n, 3))
    pi = rng.random((ni, 3)) * L
    pi[:, 2] = rng.random(ni) * L[2] * 0.5                       # every ion starts in rank 0 ...
    dz = L[2] / shape[2]
    vz = 0.9 * dz / (dt * 2.998e8)                               # ... and moves 0.9 cells up per sub-step
    vi = np.stack([np.zeros(ni), np.zeros(ni), np.full(ni, vz)], axis=1)
    ranks = []
    for r in range(world):
        s = fp.makeCylindricalParticlePusher(spec, precision="fp32")
        s.addSpecies(MP, -QE, ni if r == 0 else 300)             # rank 1 can hold 300 ions
        s.domainInit(r, world, ghost_planes=2, migrate_every=2, distributed_solve=False)
        own = np.floor(pe[:, 2] / L[2] * shape[2]).astype(int) // nzl == r
        s.domainSet(pe[own], ve[own], first_id=0 if r == 0 else int((~own).sum()), species=0)
        if r == 0:
            s.domainSet(pi, vi, first_id=0, species=1)
        ranks.append(s)
    group = fp.BoxGroup(ranks)
    group.precalc()

    def snapshot():
        return [[s.domainGet(species=sp) for sp in (0, 1)] for s in ranks]

    raised = None
    for frame in range(12):
        before = snapshot()
        try:
            group.step()
        except fp.FusionPicError as e:
            raised = str(e)
            break
    assert raised is not None and "capacity" in raised and "species 1" in raised, raised
    assert frame > 0                                            # (some migrations went through first)
    after = snapshot()
    for r in range(world):
        for sp in (0, 1):
            a, b = before[r][sp], after[r][sp]
            ia, ib = np.argsort(a["ids"]), np.argsort(b["ids"])
            assert np.array_equal(a["ids"][ia], b["ids"][ib]), (r, sp)
            assert same_bits(a["position"][ia], b["position"][ib]) and same_bits(a["velocity"][ia], b["velocity"][ib]), (r, sp)
    assert len(after[1][0]["ids"]) > 0 and len(after[1][1]["ids"]) <= 300
    for s in ranks:
        s.destroy()
