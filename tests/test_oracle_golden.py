"""The CPU oracle against the fixtures captured from the reference's own host
JavaScript (oracle/make_golden.js -> tests/golden/).  These pin everything the
reference computes on the host: constants, shader literals, the stamp, the particle
upload, texture packing, the inverse-CDF table, and the pass order / bindings of
step() and density().  The per-fragment arithmetic is NOT pinned by the reference
(it has no tests and its shaders cannot run headless): see test_oracle_physics.py.
"""
import numpy as np
import pytest

import pic_oracle as po
from helpers import load_f32gz, load_json, same_bits


def test_stamp_bits():
    g = load_json("stamp.json")
    s = po.stamp()
    assert same_bits(s, np.array(g["red"], dtype=np.float32))
    assert abs(float(s.astype(np.float64).sum()) - 1.0) < 1e-6
    assert s[60] == np.float32(0.042796533554792404)
    assert s[5 + 11 * 4] == np.float32(0.03870982676744461)
    assert s[0] == 0 and s[10] == 0 and s[110] == 0 and s[120] == 0


@pytest.mark.parametrize("key", ["demo", "squat", "c1"])
def test_constants_and_shader_literals(key):
    g = load_json("constants.json")[key]
    k = po.constants(g["spec"])
    assert k["h"] == g["u_h"]
    assert k["step_factor"] == g["u_step_factor"]
    assert g["u_pointsize"] == 11 and g["u_ratio"] == 0.01
    assert g["literal_frz_Pre1"] == [po.tofixed20(k["f_rz"])]
    assert g["literal_frz_Pre2"] == [po.tofixed20(k["f_rz"])]
    assert g["literal_fzr_Pre3"] == [po.tofixed20(k["f_zr"])] * 2
    assert g["literal_fr_fr_fz_PreA"] == [po.tofixed20(k["factor_r"]), po.tofixed20(k["factor_r"]), po.tofixed20(k["factor_z"])]
    assert g["n_programs"] == 21
    assert g["particle_count"] == g["spec"]["nparticles"] ** 2


def test_demo_constants_match_survey():
    g = load_json("constants.json")["demo"]
    assert g["u_h"] == 0.09592814371257484 and g["u_step_factor"] == 0.5996


def test_upload_and_packing_bits():
    u = load_json("upload_squat.json")
    sim = po.OracleSim(u["spec"])
    sim.set(position=u["position_in"], velocity=u["velocity_in"], E=u["E_in"], B=u["B_in"], sink_mask=u["sink_in"])
    f32 = lambda k: np.array(u[k], dtype=np.float32)
    assert same_bits(sim.pos_A, f32("position_arr")) and same_bits(sim.pos_B, f32("position_arr"))
    assert same_bits(sim.vel_A, f32("velocity_arr")) and same_bits(sim.vel_B, f32("velocity_arr"))
    assert same_bits(sim.E, f32("E_arr")) and same_bits(sim.B, f32("B_arr"))
    assert same_bits(sim.sink, f32("sink_mask_arr"))
    # 0.1 * (1/0.35) is not representable: the upload rounds the double product once
    assert sim.pos_A[0] == np.float32(0.1 * (1 / 0.35))
    # set() copies position and velocity into BOTH ping-pong buffers (empic.js:1212-1243)
    targets = [(d["program"], d["target"]) for d in u["set_draws"]]
    for t in ("position_A", "position_B", "velocity_A", "velocity_B", "E", "B", "sink_mask", "inv_cdf"):
        assert ("Set", t) in targets


@pytest.mark.parametrize("name", ["block", "ragged", "interior", "squat_random"])
def test_inverse_cdf_bits(name):
    j = load_json("inv_cdf_%s.json" % name)
    want = load_f32gz(j["file"])
    rc, got = po.inv_cdf(np.array(j["pdf"], dtype=np.float64))
    assert rc == 0
    assert same_bits(got, want)
    if "nan_count" in j:
        assert int(np.isnan(got).sum()) == j["nan_count"]


def test_inverse_cdf_quirk_q3_nan_sites():
    """inv_cdf[0,0].y is NaN when the first column of the selected row is empty."""
    j = load_json("inv_cdf_ragged.json")
    got = load_f32gz(j["file"]).reshape(512, 512, 4)
    assert np.isnan(got[0, 0, 1])
    assert not np.isnan(got[0, 0, 0])


def test_inverse_cdf_reference_throws():
    j = load_json("inv_cdf_throws.json")
    assert j["threw"] == "TypeError"
    rc, _ = po.inv_cdf(np.array(j["pdf"], dtype=np.float64))
    assert rc != 0


def test_validation_fixture_shape():
    v = load_json("validation.json")
    assert v["missing_radius"] == ".radius <- Non-optional property is undefined!"
    assert v["string_nr"] == ".nr <- Property does not match any given possible types!"


# ---- pass order and bindings of step() / density() / precalc()

PASS_FN = {"StepRandA": "rand", "StepRandB": "rand", "StepVelocityA": "vel", "StepVelocityB": "vel",
           "StepPositionA": "pos", "StepPositionB": "pos"}


def test_step_follows_reference_draw_order():
    """Replay step() pass by pass exactly as the reference issues its draws (program,
    source textures, target) with the oracle's single-pass functions, and compare with
    the oracle's own orchestration orc_f32_step."""
    d = load_json("draw_order.json")
    assert [x["program"] for x in d["step"]] == ["StepRandB", "StepVelocityB", "StepPositionB",
                                                   "StepRandA", "StepVelocityA", "StepPositionA"]
    from helpers import frame_sink, make_spec, uniform_plasma
    import ctypes
    spec = make_spec(24, 20, 30)
    n = 900
    rng = np.random.default_rng(3)
    B = rng.normal(0, 0.4, size=(24, 20, 3)); E = rng.normal(0, 1e4, size=(24, 20, 3))
    pos, vel, entropy, rand = uniform_plasma(n, spec, seed=9, v_th=0.02)
    sims = [po.OracleSim(spec), po.OracleSim(spec)]
    for s in sims:
        s.set(E=E, B=B, position=pos, velocity=vel, sink_mask=frame_sink(24, 20), source_pdf=frame_sink(24, 20))
        s.set_random_state(entropy, rand)
        s.precalc()
    a, b = sims
    a.step(1)
    tex = {"position_A": b.pos_A, "position_B": b.pos_B, "velocity_A": b.vel_A, "velocity_B": b.vel_B,
           "rand_A": b.rand_A, "rand_B": b.rand_B, "entropy_tex": b.entropy, "R1": b.R1, "R2": b.R2, "R3": b.R3,
           "A": b.A, "sink_mask": b.sink, "inv_cdf": b.inv_cdf}
    P = lambda arr: arr.ctypes.data_as(ctypes.c_void_p)
    lib = po.lib()
    for draw in d["step"]:
        r, tgt = draw["reads"], tex[draw["target"]]
        assert draw["blend"] is None and draw["triangles"] == 6
        kind = PASS_FN[draw["program"]]
        if kind == "rand":
            lib.orc_f32_step_rand(P(tex[r["u_rand"]]), P(tex[r["u_entropy"]]), P(tgt), ctypes.c_size_t(n))
        elif kind == "vel":
            lib.orc_f32_step_velocity(P(tex[r["u_position"]]), P(tex[r["u_velocity"]]), P(tex[r["u_rand"]]),
                                      P(tex[r["u_R_1"]]), P(tex[r["u_R_2"]]), P(tex[r["u_R_3"]]), P(tex[r["u_A"]]),
                                      24, 20, P(tgt), ctypes.c_size_t(n))
        else:
            # the fixture was recorded on the "squat" spec: its uniform is that spec's dt*c
            assert draw["uniforms"]["u_step_factor"] == load_json("constants.json")["squat"]["u_step_factor"]
            lib.orc_f32_step_position(P(tex[r["u_position"]]), P(tex[r["u_velocity"]]), P(tex[r["u_rand"]]),
                                      P(tex[r["u_sink"]]), P(tex[r["u_inv_cdf"]]), 24, 20,
                                      ctypes.c_float(b.step_factor), P(tgt), ctypes.c_size_t(n))
    for x, y in ((a.pos_A, b.pos_A), (a.vel_A, b.vel_A), (a.rand_A, b.rand_A), (a.pos_B, b.pos_B), (a.vel_B, b.vel_B)):
        assert same_bits(x, y)


def test_density_and_precalc_draw_order():
    d = load_json("draw_order.json")
    dens = d["density"]
    assert [x["program"] for x in dens][:4] == ["Moments01", "NormalizeMoments01", "AvgMoments", "Set"]
    m = dens[0]
    assert m["target"] == "moments01" and m["blend"] == ["ONE", "ONE"] and m["clear_color"] == [0, 0, 0, 0]
    assert m["reads"] == {"u_position": "position_A", "u_velocity": "velocity_A", "u_shape": "shape_tex"}
    assert m["points"] == 9 and m["uniforms"]["u_pointsize"] == 11
    assert dens[1]["reads"] == {"u_moments01": "moments01"} and dens[1]["target"] == "moments01_norm"
    assert dens[2]["reads"] == {"u_next": "moments01_norm", "u_avg": "moments01_avgB"} and dens[2]["target"] == "moments01_avgA"
    assert dens[3]["reads"] == {"u_value": "moments01_avgA"} and dens[3]["target"] == "moments01_avgB"
    assert [x["target"] for x in d["precalc"]] == ["R1", "R2", "R3", "A"]
    assert d["precalc"][3]["reads"] == {"u_B": "B", "u_E": "E"}
    assert all(x["target"] == "B" and x["blend"] == ["ONE", "ONE"] for x in d["painters"])
    assert d["api"] == ["addBTheta", "addBZ", "addCurrentLoop", "addCurrentZ", "addSpindleCuspPlasmaField", "canvas",
                        "density", "precalc", "set", "step"]
