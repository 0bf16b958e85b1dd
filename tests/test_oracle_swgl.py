"""The CPU restatement against the reference's own shaders evaluated in software.

tests/golden/swgl_scene.* were written by oracle/make_golden.js section 8: the reference's
factory, set(), painters, precalc(), step() and density() ran under Node with util.webGL
replaced by oracle/swgl.js, which evaluates the shader strings the reference passes to
linkProgram (oracle/glsl_eval.js, one float32 rounding per operation).  Every pass of the
hot path is compared here, bit for bit where the arithmetic is IEEE (+ - * / sqrt, lookups,
selects), within 2 ulp where cos() is involved (the current-loop painter).

This pins the oracle's TRANSCRIPTION of the shaders: operand order, swizzles, bindings,
constants, ping-pong order, blending.  It cannot pin what a GPU's GLSL compiler does with the
same text (built-in precision, contraction); DESIGN.md section 3 says so.
"""
import numpy as np
import pytest

from helpers import load_f32gz, load_json, same_bits
from pic_oracle import OracleSim


def lcg_entropy(seed, count=4 * 1024 * 1024):
    """u32 stream x <- 1664525 x + 1013904223, value = float32(x / 0xFFFFFFFF) (fixture's entropy_rule)."""
    a, c = np.uint32(1664525), np.uint32(1013904223)
    with np.errstate(over="ignore"):
        apow = np.multiply.accumulate(np.full(count, a, dtype=np.uint32), dtype=np.uint32)   # a^1 .. a^count
        geo = np.add.accumulate(np.concatenate([[np.uint32(1)], apow[:-1]]), dtype=np.uint32)  # 1 + a + .. + a^(k-1)
        x = apow * np.uint32(seed) + c * geo
    return (x.astype(np.float64) / float(0xFFFFFFFF)).astype(np.float32)


SCENES = ["swgl_scene", "swgl_tall"]


@pytest.fixture(scope="module", params=SCENES)
def scene(request):
    meta = load_json(request.param + ".json")
    blob = load_f32gz(meta["file"])
    get = lambda key: blob[meta["index"][key][0]: meta["index"][key][0] + meta["index"][key][1]]
    return meta, get


@pytest.fixture(scope="module")
def replay(scene):
    """Run the oracle through the fixture's call sequence, keeping a snapshot per stage."""
    meta, _ = scene
    sim = OracleSim(meta["spec"], np.float32)
    sim.set_random_state(entropy=lcg_entropy(meta["entropy_lcg_seed"]), rand=np.asarray(meta["rand0"], dtype=np.float32))
    sim.set(position=meta["position_in"], velocity=meta["velocity_in"], E=meta["E_in"], B=meta["B_in"],
            sink_mask=meta["sink_in"], source_pdf=meta["pdf_in"])
    snaps = {}

    def snap(stage, **arrays):
        for k, v in arrays.items():
            snaps[stage + "/" + k] = v.copy()

    snap("set", position_A=sim.pos_A, velocity_A=sim.vel_A, rand_A=sim.rand_A, E=sim.E, B=sim.B, sink_mask=sim.sink)
    for call in meta["painters"]:
        {"addCurrentLoop": sim.add_current_loop, "addCurrentZ": sim.add_current_z, "addBZ": sim.add_bz,
         "addBTheta": sim.add_btheta}[call[0]](*call[1:])
    snap("painted", E=sim.E, B=sim.B)
    sim.precalc()
    snap("precalc", R1=sim.R1, R2=sim.R2, R3=sim.R3, A=sim.A)
    for k in range(1, meta["frames"] + 1):
        sim.step()
        snap("step%d" % k, position_A=sim.pos_A, velocity_A=sim.vel_A, rand_A=sim.rand_A)
        sim.density()
        snap("density%d" % k, moments01=sim.moments, moments01_norm=sim.norm, avg=sim.avg_A)
    return snaps


def test_entropy_stream_matches_generator(scene):
    """The regenerated entropy table starts with the values the fixture's factory drew."""
    meta, _ = scene
    e = lcg_entropy(meta["entropy_lcg_seed"], 8)
    x = meta["entropy_lcg_seed"]
    want = []
    for _ in range(8):
        x = (1664525 * x + 1013904223) % 2 ** 32
        want.append(np.float32(x / 0xFFFFFFFF))
    assert same_bits(e, np.asarray(want, dtype=np.float32))


@pytest.mark.parametrize("name", ["position_A", "velocity_A", "rand_A", "E", "B", "sink_mask"])
def test_upload_pass(scene, replay, name):
    _, get = scene
    assert same_bits(replay["set/" + name], get("set/" + name))


def test_painters(scene, replay):
    """E untouched; B within 2 ulp of the loop's peak (cos in the loop shape), uniform adds exact."""
    _, get = scene
    assert same_bits(replay["painted/E"], get("painted/E"))
    want, got = get("painted/B").reshape(-1, 4), replay["painted/B"].reshape(-1, 4)
    assert same_bits(got[:, 3], want[:, 3])
    scale = np.abs(want[:, :3]).max(axis=0)
    assert np.all(np.abs(got[:, :3] - want[:, :3]) <= 2 * np.finfo(np.float32).eps * scale)


def test_precalc_from_fixture_fields(scene):
    """Boris matrix and kick from the fixture's painted fields: bit-exact (no cos involved)."""
    meta, get = scene
    sim = OracleSim(meta["spec"], np.float32)
    sim.E[:], sim.B[:] = get("painted/E"), get("painted/B")
    sim.precalc()
    for name, arr in (("R1", sim.R1), ("R2", sim.R2), ("R3", sim.R3), ("A", sim.A)):
        assert same_bits(arr, get("precalc/" + name)), name


def test_step_and_density_from_fixture_coefficients(scene):
    """Six frames of step()+density() driven by the fixture's own coefficient textures: every
    particle texel and every density texel bit-exact, including sink hits and re-injection."""
    meta, get = scene
    sim = OracleSim(meta["spec"], np.float32)
    sim.set_random_state(entropy=lcg_entropy(meta["entropy_lcg_seed"]))
    sim.set(sink_mask=meta["sink_in"], source_pdf=meta["pdf_in"])
    sim.pos_A[:], sim.vel_A[:], sim.rand_A[:] = get("set/position_A"), get("set/velocity_A"), get("set/rand_A")
    sim.R1[:], sim.R2[:], sim.R3[:], sim.A[:] = (get("precalc/" + k) for k in ("R1", "R2", "R3", "A"))
    before = sim.pos_A.copy()
    for k in range(1, meta["frames"] + 1):
        sim.step()
        for name, arr in (("position_A", sim.pos_A), ("velocity_A", sim.vel_A), ("rand_A", sim.rand_A)):
            assert same_bits(arr, get("step%d/%s" % (k, name))), (k, name)
        sim.density()
        assert same_bits(sim.moments, get("density%d/moments01" % k)), k
        assert same_bits(sim.norm, get("density%d/moments01_norm" % k)), k
        current = get("density%d/moments01_avg%s" % (k, "A"))
        assert same_bits(sim.avg_A, current) or same_bits(sim.avg_A, get("density%d/moments01_avgB" % k)), k
    after = get("step%d/position_A" % meta["frames"])
    assert not same_bits(before, after)


def test_fixture_exercises_sink_and_reinjection(scene):
    """The scene is only worth pinning if particles die, wait and come back in it."""
    meta, get = scene
    alive = [get("set/position_A").reshape(-1, 4)[:, 3] > 0.5]
    alive += [get("step%d/position_A" % k).reshape(-1, 4)[:, 3] > 0.5 for k in range(1, meta["frames"] + 1)]
    died = sum(int(np.sum(a & ~b)) for a, b in zip(alive, alive[1:]))
    reborn = sum(int(np.sum(~a & b)) for a, b in zip(alive, alive[1:]))
    assert died >= 10 and reborn >= 10, (died, reborn)


def _positions(name):
    meta = load_json(name + ".json")
    blob = load_f32gz(meta["file"])
    n = meta["spec"]["nparticles"] ** 2
    return [blob[meta["index"]["step%d/position_A" % k][0]:][:4 * n].reshape(-1, 4) for k in range(1, meta["frames"] + 1)]


def test_scenes_reach_the_edge_cases():
    """swgl_tall: particles outside the unit square (clamped lookups, points clipped from the
    deposit).  swgl_scene: a re-injection from a NaN site of the inverse CDF (quirk Q3) and a
    particle exactly at r = 0 (Q2, Q13)."""
    tall, squat = _positions("swgl_tall"), _positions("swgl_scene")
    assert any(np.nanmax(np.hypot(p[:, 0], p[:, 1])) > 1.0 for p in tall), "no particle left through the outer wall"
    assert any((p[:, 2] < 0).any() or (p[:, 2] > 1).any() for p in tall), "no particle left through an end wall"
    assert any(np.isnan(p[:, :3]).any() for p in squat), "no NaN re-injection site was hit"
    assert any((np.hypot(p[:, 0], p[:, 1]) == 0).any() for p in squat), "no particle at r = 0"
