"""Analytic and property tests of the CPU oracle's per-fragment arithmetic.

The reference pins none of this (no tests; shaders cannot run headless), so these
known-answer tests are what anchors the restatement: Boris phase advance, orthogonality
of R, NGP indexing at cell edges, the re-injection path, the RNG recurrences, the
single-particle stamp, normalisation and EMA, and float-vs-double agreement.
"""
import ctypes

import numpy as np
import pytest
from hypothesis import given, settings
from hypothesis import strategies as st

import pic_oracle as po
from helpers import frame_sink, make_spec, uniform_plasma


def P(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def uniform_bz_sim(bz, dtype=np.float64, n=1, nr=16, nz=16, radius=1.0, height=1.0):
    spec = make_spec(nr, nz, 1, radius=radius, height=height)
    sim = po.OracleSim(spec, dtype=dtype, count=n)
    B = np.zeros((nr, nz, 3)); B[..., 2] = bz
    sim.set(B=B, sink_mask=np.ones((nr, nz)), source_pdf=np.ones((nr, nz)))
    sim.precalc()
    return sim, spec


def test_boris_rotation_matrix_is_orthogonal_when_isotropic():
    """R = (1 - h^2B^2 f) I + f h^2 B B^T + f h [xB] is a rotation when factor_r == factor_z."""
    spec = make_spec(8, 8, 1)
    sim = po.OracleSim(spec, dtype=np.float64)
    rng = np.random.default_rng(0)
    sim.set(B=rng.normal(0, 2.0, size=(8, 8, 3)))
    sim.precalc()
    for c in range(64):
        R = np.stack([sim.R1[4 * c:4 * c + 3], sim.R2[4 * c:4 * c + 3], sim.R3[4 * c:4 * c + 3]])
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-12)
        assert abs(np.linalg.det(R) - 1) < 1e-12
    assert np.all(sim.A.reshape(-1, 4)[:, :3] == 0)  # E = 0


def test_boris_rotation_angle():
    bz = 0.7
    sim, spec = uniform_bz_sim(bz)
    h = sim.k["h"]
    R = np.stack([sim.R1[:3], sim.R2[:3], sim.R3[:3]])
    theta = 2 * np.arctan(h * bz)
    assert abs(R[0, 0] - np.cos(theta)) < 1e-14 and abs(R[0, 1] - np.sin(theta)) < 1e-14
    assert abs(R[2, 2] - 1) < 1e-14


def test_anisotropy_scaling_of_R():
    """Off-axis terms carry factor_r/factor_z resp. factor_z/factor_r (empic.js:527, :566, :606)."""
    spec = make_spec(4, 4, 1, radius=0.5, height=2.0)
    iso = make_spec(4, 4, 1, radius=1.0, height=1.0)
    B = np.random.default_rng(1).normal(0, 1, size=(4, 4, 3))
    a, b = po.OracleSim(spec, dtype=np.float64), po.OracleSim(iso, dtype=np.float64)
    for s in (a, b):
        s.set(B=B); s.precalc()
    frz = (1 / 0.5) / (1 / 2.0)
    assert np.allclose(a.R1[2::4], b.R1[2::4] * frz) and np.allclose(a.R2[2::4], b.R2[2::4] * frz)
    assert np.allclose(a.R3[0::4], b.R3[0::4] / frz) and np.allclose(a.R3[1::4], b.R3[1::4] / frz)
    assert np.allclose(a.R1[0::4], b.R1[0::4]) and np.allclose(a.R3[2::4], b.R3[2::4])


def test_precalc_quirk_q1_formula():
    """A per the reference: scalar h*(E.B) added to each component (empic.js:645)."""
    spec = make_spec(2, 2, 1)
    E = np.zeros((2, 2, 3)); B = np.zeros((2, 2, 3))
    E[..., 0] = 1e5; B[..., 0] = 0.5; B[..., 2] = 0.25
    q, p = po.OracleSim(spec, dtype=np.float64), po.OracleSim(spec, dtype=np.float64, physical_a=True)
    for s in (q, p):
        s.set(E=E, B=B); s.precalc()
    h = q.k["h"]; hB2 = h * h * (0.5 ** 2 + 0.25 ** 2); f = 2 / (1 + hB2)
    a, b = h * (2 - hB2 * f), h * h * f
    cross = np.array([0.0, 0 * 0.5 - 1e5 * 0.25, 0.0])
    dot = 1e5 * 0.5
    want_q = (a * E[0, 0] + b * (cross + h * dot)) / 2.998e8
    want_p = (a * E[0, 0] + b * (cross + h * dot * B[0, 0])) / 2.998e8
    assert np.allclose(q.A[:3], want_q, rtol=1e-13) and np.allclose(p.A[:3], want_p, rtol=1e-13)
    assert not np.allclose(q.A[:3], p.A[:3])


def test_single_particle_gyration_closes():
    """Leap-frog in uniform Bz: |v| conserved, phase advance 2 atan(hB) per sub-step,
    gyro-centre stays put."""
    bz = 0.05
    sim, spec = uniform_bz_sim(bz, nr=64, nz=64)
    v0 = np.array([2e-4, 0.0, 1e-5])
    sim.set(position=[[0.5, 0.1, 0.5]], velocity=[v0])
    h = sim.k["h"]
    theta = 2 * np.arctan(h * bz)
    speeds, angles = [], []
    for _ in range(50):
        sim.step()
        v = sim.velocities()[0]
        speeds.append(np.hypot(v[0], v[1])); angles.append(np.arctan2(v[1], v[0]))
    assert np.allclose(speeds, np.hypot(v0[0], v0[1]), rtol=1e-12)
    d = np.diff(np.unwrap(angles))
    assert np.allclose(np.abs(d), 2 * theta, rtol=1e-9)
    assert sim.velocities()[0][2] == v0[2]
    assert sim.alive()[0] == 1


def test_ngp_at_cell_edges_and_clamps():
    spec = make_spec(8, 4, 1)
    sim = po.OracleSim(spec, count=6)
    pos = np.array([[0.125, 0, 0.25], [0.1249999, 0, 0.2499999], [0.0, 1e-9, 0.0], [0.999999, 0, 0.999999],
                    [1.5, 0, 2.0], [0.3, 0, -0.5]])
    sim.set(position=pos)
    cells = sim.cells()
    assert list(cells) == [1 + 8 * 1, 0 + 8 * 0, 0, 7 + 8 * 3, 7 + 8 * 3, 2 + 8 * 0]


def test_reinjection_path_and_reseed_velocity():
    """A particle stepping onto a sink cell is replaced by inv_cdf[NGP(u1,u2)] with y = 0 and
    alive = 0; on the next sub-step its velocity is 0.001*(2*rand.xyz - 1) of THAT
    sub-step's random state (empic.js:719, :772)."""
    spec = make_spec(8, 8, 1)
    sim = po.OracleSim(spec, dtype=np.float32, count=1)
    sink = np.ones((8, 8)); sink[7, :] = 0
    pdf = np.zeros((8, 8)); pdf[0:3, 3] = 1.0   # row 0 must carry weight or the reference's set() throws
    sim.set(position=[[0.86, 0.0, 0.5]], velocity=[[0.1, 0.0, 0.0]], sink_mask=sink, source_pdf=pdf)
    rand0 = np.array([[0.25, 0.75, 0.3, 0.6]], dtype=np.float32)
    entropy = np.random.default_rng(5).random(4 * 1024 * 1024, dtype=np.float32)
    sim.set_random_state(entropy, rand0)
    n = ctypes.c_size_t(1)
    lib = po.lib()
    # sub-step 1 by hand
    lib.orc_f32_step_rand(P(sim.rand_A), P(sim.entropy), P(sim.rand_B), n)
    lib.orc_f32_step_velocity(P(sim.pos_A), P(sim.vel_A), P(sim.rand_A), P(sim.R1), P(sim.R2), P(sim.R3), P(sim.A), 8, 8, P(sim.vel_B), n)
    lib.orc_f32_step_position(P(sim.pos_A), P(sim.vel_B), P(sim.rand_A), P(sim.sink), P(sim.inv_cdf), 8, 8,
                              ctypes.c_float(sim.step_factor), P(sim.pos_B), n)
    assert np.all(sim.vel_B[:3] == 0)            # coefficients are zero before precalc
    # without precalc v' = 0, so push it over the edge by hand instead
    sim.vel_B[0] = 0.3
    lib.orc_f32_step_position(P(sim.pos_A), P(sim.vel_B), P(sim.rand_A), P(sim.sink), P(sim.inv_cdf), 8, 8,
                              ctypes.c_float(sim.step_factor), P(sim.pos_B), n)
    t = 4 * (int(0.25 * 512) + 512 * int(0.75 * 512))
    assert sim.pos_B[3] == 0 and sim.pos_B[1] == 0
    assert sim.pos_B[0] == sim.inv_cdf[t] and sim.pos_B[2] == sim.inv_cdf[t + 1]
    assert 0 <= sim.pos_B[0] <= 3 / 8 and 3 / 8 <= sim.pos_B[2] <= 4 / 8
    # sub-step 2: velocity pass sees alive = 0 and the advanced random state rand_B
    lib.orc_f32_step_velocity(P(sim.pos_B), P(sim.vel_B), P(sim.rand_B), P(sim.R1), P(sim.R2), P(sim.R3), P(sim.A), 8, 8, P(sim.vel_A), n)
    want = np.float32(0.001) * (np.float32(2) * sim.rand_B[:3] - np.float32(1))
    assert np.array_equal(sim.vel_A[:3], want)


def test_rng_recurrences():
    """u <- wrap(u + s.xy) with m > 1 ? m-1 : m (m == 1 stays 1, quirk Q5); c <- 4x(1-x),
    x = 0.999c + 0.001 s.zw (empic.js:800-807)."""
    ent = np.zeros(4 * 1024 * 1024, dtype=np.float32)
    t = 4 * (int(np.float32(0.5) * 1024) + 1024 * int(np.float32(0.25) * 1024))
    ent[t:t + 4] = [0.75, 0.5, 0.125, 1.0]
    rin = np.array([0.5, 0.5, 0.5, 0.25], dtype=np.float32)
    out = np.zeros(4, dtype=np.float32)
    po.lib().orc_f32_step_rand(P(rin), P(ent), P(out), ctypes.c_size_t(1))
    f = np.float32
    assert out[0] == f(f(0.5) + f(0.75)) - f(1)
    assert out[1] == f(1.0)                       # 0.5 + 0.5 == 1 is not > 1
    x0 = f(f(0.999) * f(0.5)) + f(f(0.001) * f(0.125))
    assert out[2] == f(f(4) * x0) * (f(1) - x0)
    x1 = f(f(0.999) * f(0.25)) + f(f(0.001) * f(1.0))
    assert out[3] == f(f(4) * x1) * (f(1) - x1)


def test_single_particle_deposit_is_the_stamp():
    spec = make_spec(20, 24, 1)
    sim = po.OracleSim(spec, dtype=np.float64, count=1)
    sim.set(position=[[0.3, 0.4, 0.5]], velocity=[[3e-4, 0.0, -1e-4]])
    sim.deposit()
    m = sim.moments.reshape(24, 20, 4)
    ic, jc = int(0.5 * 20), int(0.5 * 24)
    stamp = po.stamp().astype(np.float64).reshape(11, 11)
    assert np.allclose(m[jc - 5:jc + 6, ic - 5:ic + 6, 3], 0.001 * stamp[::-1], rtol=1e-12)
    assert abs(m[..., 3].sum() - 0.001) < 1e-9
    vr = 3e-4 * 0.6          # direction (0.6, 0.8)
    vt = -3e-4 * 0.8
    assert np.allclose(m[jc, ic, :3], 0.001 * stamp[5, 5] * np.array([vr, vt, -1e-4]), rtol=1e-9)
    total = m.sum(axis=(0, 1))
    assert np.allclose(total[:3], 0.001 * np.array([vr, vt, -1e-4]), rtol=1e-6)


def test_deposit_crops_at_edges_and_clips_outside():
    spec = make_spec(16, 16, 1)
    sim = po.OracleSim(spec, dtype=np.float64, count=3)
    sim.set(position=[[0.01, 0.0, 0.01], [1.2, 0.0, 0.5], [0.5, 0.0, 1.01]], velocity=np.zeros((3, 3)))
    sim.deposit()
    m = sim.moments.reshape(16, 16, 4)[..., 3]
    stamp = po.stamp().astype(np.float64).reshape(11, 11)
    assert np.allclose(m[:6, :6], 0.001 * stamp[::-1][5:, 5:])
    assert m[6:, :].sum() == 0 and m[:, 6:].sum() == 0
    assert list(sim.deposit_cells()) == [0, -1, -1]


def test_normalise_and_ema():
    spec = make_spec(4, 2, 1)
    sim = po.OracleSim(spec, dtype=np.float64, count=1)
    sim.moments[:] = 0
    sim.moments[4 * 5:4 * 5 + 4] = [2e-4, -4e-4, 6e-4, 2e-3]      # cell i=1, j=1
    sim.density_finish()
    x = (1 + 0.5) / 4
    assert np.allclose(sim.norm[20:24], np.array([0.1, -0.2, 0.3, 2e-3]) * 1000 * 0.5 / x)
    assert np.all(sim.norm[:20] == 0)
    assert np.allclose(sim.avg_A[20:24], 0.01 * sim.norm[20:24])
    first = sim.avg_A.copy()
    sim.density_finish()
    assert np.allclose(sim.avg_A[20:24], 0.01 * sim.norm[20:24] + 0.99 * first[20:24])
    assert np.array_equal(sim.avg_A, sim.avg_B)


def test_float_and_double_oracles_agree_within_tolerance():
    spec = make_spec(48, 40, 60, radius=1.0, height=2.0)
    n = 3600
    rng = np.random.default_rng(2)
    B = rng.normal(0, 0.3, size=(48, 40, 3)); B[..., 2] += 1
    E = rng.normal(0, 1e4, size=(48, 40, 3))
    pos, vel, entropy, rand = uniform_plasma(n, spec, seed=4, margin=0.1)
    sims = [po.OracleSim(spec, dtype=np.float32), po.OracleSim(spec, dtype=np.float64)]
    for s in sims:
        s.set(E=E, B=B, position=pos, velocity=vel, sink_mask=np.ones((48, 40)), source_pdf=np.ones((48, 40)))
        s.set_random_state(entropy, rand)
        s.precalc(); s.step()
    a, b = sims
    assert np.allclose(a.positions(), b.positions(), rtol=1e-5, atol=1e-6)
    assert np.allclose(a.velocities(), b.velocities(), rtol=1e-3, atol=1e-8)
    assert np.mean(a.cells() == b.cells()) > 0.995


def test_particle_count_is_constant_and_lost_particles_return():
    spec = make_spec(16, 16, 40, radius=0.3, height=0.3)
    n = 1600
    pos, vel, entropy, rand = uniform_plasma(n, spec, seed=8, v_th=0.05)
    sim = po.OracleSim(spec)
    sim.set(position=pos, velocity=vel, sink_mask=frame_sink(16, 16), source_pdf=frame_sink(16, 16))
    sim.set_random_state(entropy, rand)
    B = np.zeros((16, 16, 3)); B[..., 2] = 0.01
    sim.set(B=B); sim.precalc()
    dead_seen = 0
    for _ in range(10):
        sim.step()
        assert sim.positions().shape == (n, 3)
        dead_seen += int((sim.alive() == 0).sum())
        sim.deposit()
        inside = int((sim.deposit_cells() >= 0).sum())
        assert abs(sim.moments.reshape(-1, 4)[:, 3].astype(np.float64).sum() - 0.001 * inside) < 0.001 * inside * 0.2
    assert dead_seen > 0


@settings(max_examples=20, deadline=None)
@given(st.integers(min_value=0, max_value=2 ** 31 - 1))
def test_deposit_is_permutation_invariant_within_rounding(seed):
    """Additive blending has no defined order in the reference; any order must agree to rounding."""
    spec = make_spec(24, 24, 1)
    n = 500
    rng = np.random.default_rng(seed)
    pos, vel, _, _ = uniform_plasma(n, spec, seed=seed)
    a, b = po.OracleSim(spec, count=n), po.OracleSim(spec, count=n)
    perm = rng.permutation(n)
    a.set(position=pos, velocity=vel); b.set(position=pos[perm], velocity=vel[perm])
    a.deposit(); b.deposit()
    scale = np.abs(a.moments).max()
    assert np.abs(a.moments - b.moments).max() <= 1e-5 * scale


def test_uniform_painters():
    spec = make_spec(8, 4, 1, radius=0.5, height=2.0)
    sim = po.OracleSim(spec, dtype=np.float64)
    sim.add_bz(0.25); sim.add_btheta(-0.5); sim.add_current_z(1e5)
    B = sim.B.reshape(4, 8, 4)
    assert np.all(B[..., 2] == 0.25) and np.all(B[..., 3] == 3)
    tx = (np.arange(8) + 0.5) / 8
    assert np.allclose(B[0, :, 1], -0.5 + 1e5 * 1.25663706e-6 / (2 * 3.14159265359 * tx))


def test_current_loop_on_axis_field():
    """Bz on the axis in the loop's plane.  The reference's constant is u_R*0.001*mu0/4pi
    summed over 1000 segments of HALF the circle (empic.js:313-324): the segment length
    0.001*R stands where pi*R/1000 (x2 halves) belongs, so its field is the physical
    mu0 I / (2R) divided by 2*pi.  The restatement keeps that."""
    spec = make_spec(100, 100, 1, radius=1.0, height=1.0)
    sim = po.OracleSim(spec, dtype=np.float64)
    R, I = 0.5, 1e6
    sim.add_current_loop(R, 0.5, I)
    B = sim.B.reshape(100, 100, 4)
    bz_axis = B[50, 0, 2]
    want = 1.25663706e-6 * I / (2 * R) / (2 * np.pi)
    assert abs(bz_axis - want) / want < 0.01


def test_cic_deposit_is_bilinear_and_conserves_the_count():
    """extension shape:'cic' (no reference counterpart): weights are linear in the offset from the cell centres,
    sum to one away from the edges, are cropped at the edges like the reference's sprite"""
    import pic_oracle as po
    from helpers import make_spec
    spec = make_spec(16, 12, 1, radius=1.0, height=1.0)
    sim = po.OracleSim(spec, dtype=np.float64, count=1, shape="cic")
    for r, z in ((0.40625, 0.5), (0.42, 0.37), (0.5 / 16 * 0.3, 0.5), (0.999, 0.999)):
        sim.set(position=[[r, 0.0, z]], velocity=[[1e-3, 2e-3, 3e-3]])
        sim.deposit()
        m = sim.moments.reshape(12, 16, 4)
        gi, gj = r * 16 - 0.5, z * 12 - 0.5
        i0, j0 = int(np.floor(gi)), int(np.floor(gj))
        want = np.zeros((12, 16))
        for b, wz in ((0, 1 - (gj - j0)), (1, gj - j0)):
            for a, wr in ((0, 1 - (gi - i0)), (1, gi - i0)):
                if 0 <= i0 + a < 16 and 0 <= j0 + b < 12:
                    want[j0 + b, i0 + a] = wr * wz
        assert np.allclose(m[..., 3], 0.001 * want, rtol=1e-12, atol=1e-18)
        assert np.allclose(m[..., 0], 0.001 * 1e-3 * want, rtol=1e-12, atol=1e-20)      # v_r = v_x on the plane y = 0
    interior = want.sum()
    assert interior < 1.0                                                              # the last point is cropped at the corner
