"""Known-answer tests of the full-EM mode of the CART3D oracle (solver 'yee'; BASELINE configs[4]).

PARITY UNPINNED (no reference counterpart).  Anchors ("Extension known-answers"): the lattice
continuity equation d(rho)/dt + div J = 0 — here EXACT in integers, because charge and current are deposited in fixed
point from the same quantised positions; the vacuum dispersion of the Yee scheme; Gauss's law kept to rounding by the
charge-conserving current; the energy budget; the electrostatic limit (omega_p).
"""
import numpy as np
import pytest

import es3d_oracle as eo

ME, QE = 9.109e-31, -1.602e-19
C = eo.SPEED_OF_LIGHT


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


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_continuity_holds_exactly_in_integers(dtype):
    """96 (rho_fixed^(n+1) - rho_fixed^n) + div J_fixed = 0 at every node, every step: fast particles that cross cell
    faces in one, two and three directions at once, both species, positions on faces and nodes, wrap at the box edges"""
    rng = np.random.default_rng(2)
    shape, L = (8, 6, 10), (0.008, 0.006, 0.010)
    n = 3000
    sim = eo.OracleES3D(em_spec(shape, L, n, cfl_dt(shape, L), macro_weight=1e6), dtype)
    ions = sim.add_species(1836 * ME, -2 * QE, 500)
    pos = rng.random((n, 3)) * L
    pos[:6] = [[0, 0, 0], [L[0], L[1], L[2]], [1e-3, 2e-3, 3e-3], [L[0] - 1e-12, 0, 5e-3], [4e-3, 3e-3, 0.0], [7.9999e-3, 5.9999e-3, 9.9999e-3]]
    sim.set(position=pos, velocity=rng.normal(0, 0.35, (n, 3)))        # up to a cell per step: every crossing pattern occurs
    sim.set(position=rng.random((500, 3)) * L, velocity=rng.normal(0, 0.1, (500, 3)), species=ions)
    sim.add_b(0.01, -0.02, 0.03)
    sim.precalc()
    for step in range(4):
        before = sim.rho_fixed.copy()
        sim.em_substep()
        sim.deposit()
        res = 96 * (sim.rho_fixed - before).reshape(shape[2], shape[1], shape[0]) + divergence(sim.J_fixed, shape)
        assert np.count_nonzero(sim.J_fixed) > 1000 and not res.any(), step


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("mode", [(1, 0, 0), (2, 3, 0), (1, 2, 3)])
def test_vacuum_standing_wave_follows_the_yee_dispersion(dtype, mode):
    """E(t=0) = e0 cos(k.x) on the lattice (e0 perpendicular to k in the lattice sense), B = 0: E^n = E^0 cos(n omega dt)
    with sin^2(omega dt / 2) = (c dt)^2 sum_a sin^2(k_a d_a / 2) / d_a^2"""
    shape, L = (16, 12, 20), (0.16, 0.12, 0.2)
    dt = cfl_dt(shape, L, 0.7)
    sim = eo.OracleES3D(em_spec(shape, L, 1, dt, macro_weight=1e-30), dtype)
    sim.set(position=[[0.01, 0.01, 0.01]], velocity=[[0, 0, 0]])
    d = [L[a] / shape[a] for a in range(3)]
    i, j, k = np.meshgrid(np.arange(shape[0]), np.arange(shape[1]), np.arange(shape[2]), indexing="ij")
    idx = (i, j, k)
    kk = [2 * np.pi * mode[a] / L[a] for a in range(3)]
    # lattice-transverse polarisation: sum_a e_a * (2/d_a) sin(k_a d_a / 2) = 0, each component sampled on its own edge
    K = np.array([2 / d[a] * np.sin(kk[a] * d[a] / 2) for a in range(3)])
    e0 = np.cross(K, [0.3, -0.5, 0.8]) if np.linalg.norm(np.cross(K, [0.3, -0.5, 0.8])) > 0 else np.array([0, 1.0, 0])
    e0 = e0 / np.linalg.norm(e0) * 1e3
    E = np.zeros(shape + (3,))
    for a in range(3):
        phase = sum(kk[b] * d[b] * (idx[b] + (0.5 if b == a else 0.0)) for b in range(3))
        E[..., a] = e0[a] * np.cos(phase)
    sim.set_lattice(E=E, B=np.zeros(shape + (3,)))
    omega = 2 / dt * np.arcsin(C * dt * np.sqrt(sum((np.sin(kk[a] * d[a] / 2) / d[a]) ** 2 for a in range(3))))
    steps = 40
    for _ in range(steps):
        sim.em_substep()
    got, _ = sim.lattice_fields()
    tol = 2e-5 if dtype == np.float32 else 1e-11
    assert np.abs(got - E * np.cos(steps * omega * dt)).max() <= tol * 1e3 * steps


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_gauss_law_is_kept_and_energy_is_conserved(dtype):
    """a warm plasma started from its electrostatic field: div E - rho/eps0 stays at rounding level (the current is charge
    conserving), total energy (field + kinetic) drifts by well under a percent over 60 steps"""
    rng = np.random.default_rng(9)
    shape, L = (12, 12, 12), (0.012, 0.012, 0.012)
    n = 12 * 12 * 12 * 8
    dt = cfl_dt(shape, L, 0.5)
    wp = 0.02 / dt                        # Debye length 0.7 cells at v_th = 0.05 c: the grid heating stays small
    dens = wp ** 2 * eo.EPS0 * ME / QE ** 2
    sim = eo.OracleES3D(em_spec(shape, L, n, dt, macro_weight=dens * np.prod(L) / n), dtype)
    sim.set(position=rng.random((n, 3)) * L, velocity=rng.normal(0, 0.05, (n, 3)))
    sim.precalc()
    res, scale = sim.gauss_residual()
    tol = 3e-5 if dtype == np.float32 else 1e-11
    assert np.abs(res).max() <= tol * scale
    e0 = sim.em_field_energy() + sim.kinetic_energy()
    for _ in range(60):
        sim.em_substep()
    res, scale = sim.gauss_residual()
    assert np.abs(res).max() <= 60 * tol * scale
    e1 = sim.em_field_energy() + sim.kinetic_energy()
    assert abs(e1 / e0 - 1) < 1e-2
    assert sim.em_field_energy() > 0


def test_em_mode_reproduces_the_plasma_oscillation():
    """the electrostatic limit inside the EM scheme: cold lattice of electrons, sinusoidal velocity perturbation,
    field energy oscillates at 2 omega_p (within 2 %)"""
    nx, L, per_cell = 32, 0.32, 4
    shape, box = (nx, 4, 4), (L, L / 8, L / 8)
    n = nx * per_cell * 4 * 4
    dt = cfl_dt(shape, box, 0.5)
    wp = 0.05 / dt
    dens = wp ** 2 * eo.EPS0 * ME / QE ** 2
    sim = eo.OracleES3D(em_spec(shape, box, n, dt, macro_weight=dens * np.prod(box) / n), np.float64)
    xs = (np.arange(nx * per_cell) + 0.5) / (nx * per_cell) * L
    ys = (np.arange(4) + 0.5) / 4 * (L / 8)
    X, Y, Z = np.meshgrid(xs, ys, ys, indexing="ij")
    pos = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    vel = np.zeros_like(pos)
    vel[:, 0] = 2e-3 * np.sin(2 * np.pi / L * pos[:, 0])
    sim.set(position=pos, velocity=vel)
    sim.precalc()
    energy = []
    for _ in range(int(2.2 * 2 * np.pi / 0.05)):
        sim.em_substep()
        energy.append(sim.em_field_energy())
    energy = np.array(energy)
    t = (np.arange(len(energy)) + 1) * dt
    peaks = [i for i in range(1, len(energy) - 1) if energy[i] > energy[i - 1] and energy[i] >= energy[i + 1]]
    assert len(peaks) >= 3

    def vertex(i):
        y0, y1, y2 = energy[i - 1], energy[i], energy[i + 1]
        return t[i] + 0.5 * (y0 - y2) / (y0 - 2 * y1 + y2) * dt
    omega = np.pi / np.mean(np.diff([vertex(i) for i in peaks]))
    assert abs(omega / wp - 1) < 0.02
