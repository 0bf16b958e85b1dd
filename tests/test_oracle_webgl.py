"""The CPU restatement against the REFERENCE ITSELF executed by a real WebGL implementation.

tests/golden/webgl_* were written by oracle/make_golden_webgl.py: the reference's unmodified utilities.js / spindle.js /
empic.js / matrix_webgl.js ran in the headless Chromium of the `kaleido` package (WebGL 1 on ANGLE/SwiftShader: a real
GLSL compiler and a real rasteriser), and every frame buffer was read back with the reference's own readPixels after
set(), the painters, precalc() and every step() / density().  These are outputs of the reference, not of anything this
repository wrote; they pin the oracle's arithmetic, which the swgl_* fixtures (our own evaluator of the shader text) could
not.

What holds, and is asserted here:
  bit for bit   the upload, the inverse-CDF table (NaN sites included), the stamp, E, the uniform painters, precalc()
                from given fields (R1, R2, R3, A), every position / velocity / random-state texel and alive flag of every
                frame from given coefficients, the dense solver's every texture and returned number, and the deposit
                moments01 under the rasterised convention (oracle deposit_raster: 4 sub-pixel bits, y down, cropped
                footprints) for every texel above 1e-30 of the image's maximum;
  to tolerance  the current-loop painter (1000 cos() per cell: SwiftShader's cos is a polynomial), the normalised
                density and its running average (SwiftShader's division is not correctly rounded: <= 2 ulp observed).
"""
import hashlib

import numpy as np
import pytest

from helpers import load_f32gz, load_json, load_webgl, same_bits
from pic_oracle import OracleSim, stamp as oracle_stamp
from test_oracle_swgl import lcg_entropy

RANDOM = ["webgl_rand%d" % k for k in range(16)]      # everything drawn at random: grid, cylinder, dt, species, fields, masks, painters
SCENES = ["webgl_scene", "webgl_tall", "webgl_efield", "webgl_nan", "webgl_probe"] + RANDOM
BITS = 4    # gl.getParameter(SUBPIXEL_BITS) of the implementation that wrote the fixtures (webgl_info.json)
FLOOR = 1e-30   # relative to the image's maximum: below it live the stamp's outermost ring (<= 1.7e-34) and flushed denormals


def load_scene(name):
    meta, get, inputs = load_webgl(name)
    meta["_inputs"] = inputs
    return meta, get


@pytest.fixture(scope="module", params=SCENES)
def scene(request):
    return load_scene(request.param)


def fresh(meta, **kw):
    sim = OracleSim(meta["spec"], np.float32, **kw)
    i = meta["_inputs"]
    sim.set_random_state(entropy=lcg_entropy(meta["entropy_lcg_seed"]), rand=i["rand0"])
    sim.set(position=i["position"], velocity=i["velocity"], E=i["E"], B=i["B"], sink_mask=i["sink_mask"], source_pdf=i["source_pdf"])
    return sim


def sha(a):
    a = np.ascontiguousarray(a, dtype="<f4").copy()
    u = a.view("<u4")
    u[np.isnan(a)] = 0x7FC00000
    return hashlib.sha256(u.tobytes()).hexdigest()


PAINT = {"addCurrentLoop": "add_current_loop", "addCurrentZ": "add_current_z", "addBZ": "add_bz", "addBTheta": "add_btheta"}


def test_the_fixtures_come_from_a_real_webgl():
    info = load_json("webgl_info.json")
    assert info["have_webgl"] and info["version"].startswith("WebGL 1.0")
    assert info["OES_texture_float"] and info["WEBGL_color_buffer_float"] and info["EXT_float_blend"]
    assert info["highp_fragment"] == [127, 127, 23]
    assert info["subpixel_bits"] == BITS
    for name in SCENES + ["webgl_demo"]:
        meta = load_json(name + ".json")
        assert meta["gl"]["unmasked_renderer"] == info["unmasked_renderer"]
        assert meta["inv_cdf_fbo_equals_host_table"] and meta["inv_cdf_zw_all_zero"] and meta["stamp_gba_equal_red"]
        assert sorted(meta["api"]) == sorted(["addBTheta", "addBZ", "addCurrentLoop", "addCurrentZ", "addSpindleCuspPlasmaField",
                                               "canvas", "density", "precalc", "set", "step"])


def test_random_state_follows_the_stated_rule(scene):
    meta, get = scene
    n = meta["spec"]["nparticles"] ** 2
    rand0 = meta["_inputs"]["rand0"]
    x, want = meta["entropy_lcg_seed"], []
    for _ in range(4 * 1024 * 1024 + 8):
        x = (1664525 * x + 1013904223) & 0xFFFFFFFF
        if len(want) < 8 and _ >= 4 * 1024 * 1024:
            want.append(np.float32(x / 4294967296.0))
    assert same_bits(np.asarray(rand0[:8], dtype=np.float32), np.asarray(want))
    assert len(rand0) == 4 * n
    assert same_bits(get("set/rand_A"), np.asarray(rand0, dtype=np.float32))


@pytest.mark.parametrize("name", ["position_A", "velocity_A", "E", "B", "sink_mask"])
def test_upload(scene, name):
    meta, get = scene
    sim = fresh(meta)
    got = {"position_A": sim.pos_A, "velocity_A": sim.vel_A, "E": sim.E, "B": sim.B, "sink_mask": sim.sink}[name]
    assert same_bits(got, get("set/" + name))


def test_inverse_cdf_and_stamp(scene):
    meta, get = scene
    sim = fresh(meta)
    if "sha256" in meta:      # the compact fixtures keep the 512 x 512 table by digest
        assert sha(sim.inv_cdf.reshape(-1, 4)[:, :2]) == meta["sha256"]["set/inv_cdf_xy"]
        return
    assert same_bits(sim.inv_cdf.reshape(-1, 4)[:, :2].ravel(), get("set/inv_cdf_xy"))
    assert same_bits(oracle_stamp(), get("init/stamp_red"))


def test_painters(scene):
    """E untouched, alpha exact; the loop painter sums 1000 segments with cos() per cell and divides by r^3: SwiftShader's
    cos is a polynomial, so texels near the wire move by up to a few 1e-5 of the peak; elsewhere parts in 1e7."""
    meta, get = scene
    sim = fresh(meta)
    for call in meta["painters"]:
        getattr(sim, PAINT[call[0]])(*call[1:])
    assert same_bits(sim.E, get("painted/E"))
    want, got = get("painted/B").reshape(-1, 4), sim.B.reshape(-1, 4)
    assert same_bits(got[:, 3], want[:, 3])
    if not any(c[0] == "addCurrentLoop" for c in meta["painters"]):
        # uniform painters: adds, and in addCurrentZ one division by the varying radius (SwiftShader: within an ulp)
        assert np.all(np.abs(got - want) <= np.spacing(np.abs(want).max(axis=0)))      # an ulp of the channel's largest term
        return
    peak = np.abs(want[:, :3]).max()
    err = np.abs(got[:, :3] - want[:, :3]).max(axis=1) / peak
    assert err.max() <= 1e-4 and np.median(err) <= 1e-6, (err.max(), np.median(err))


def test_precalc_from_fixture_fields(scene):
    meta, get = scene
    sim = OracleSim(meta["spec"], np.float32)
    sim.E[:], sim.B[:] = get("painted/E"), get("painted/B")
    sim.precalc()
    for name, arr in (("R1", sim.R1), ("R2", sim.R2), ("R3", sim.R3), ("A", sim.A)):
        assert same_bits(arr, get("precalc/" + name)), name


def replay(meta, get, **kw):
    sim = fresh(meta, **kw)
    sim.R1[:], sim.R2[:], sim.R3[:], sim.A[:] = (get("precalc/" + k) for k in ("R1", "R2", "R3", "A"))
    return sim


def test_push_from_fixture_coefficients_is_bit_exact(scene):
    """K1 + K2 + K3 of every frame, sink hits and re-injections (NaN sites included) as they come."""
    meta, get = scene
    sim = replay(meta, get)
    for k in range(1, meta["frames"] + 1):
        sim.step()
        for name, arr in (("position_A", sim.pos_A), ("velocity_A", sim.vel_A), ("rand_A", sim.rand_A)):
            assert same_bits(arr, get("step%d/%s" % (k, name))), (k, name)


def test_deposit_under_the_rasterised_convention(scene):
    """moments01 of every frame: bit for bit above FLOOR, within FLOOR below it (see the header of deposit_raster)."""
    meta, get = scene
    sim = replay(meta, get, raster_bits=BITS)
    for k in range(1, meta["frames"] + 1):
        sim.step()
        sim.deposit()
        want = get("density%d/moments01" % k)
        top = np.nanmax(np.abs(want))
        big = np.abs(want) > FLOOR * top
        assert big.sum() > 100
        assert same_bits(sim.moments[big], want[big]), k
        assert np.array_equal(np.isnan(sim.moments), np.isnan(want))
        assert np.nanmax(np.abs(sim.moments - want)) <= FLOOR * top, k


def test_normalised_density_and_running_average(scene):
    """K5 divides by the count channel and by the cell's radius; SwiftShader's division is not correctly rounded, so these
    are tolerance tests (4 ulp of the texel, plus FLOOR).  A cell whose count is below FLOOR (only the stamp's outermost
    ring reached it) is excluded: there the quotient of two such numbers is whatever flushing left of them."""
    meta, get = scene
    sim = replay(meta, get, raster_bits=BITS)
    masked = np.zeros(meta["spec"]["nr"] * meta["spec"]["nz"], dtype=bool)
    for k in range(1, meta["frames"] + 1):
        sim.step()
        sim.density()
        m = get("density%d/moments01" % k).reshape(-1, 4)
        masked |= (np.abs(m[:, 3]) <= FLOOR) & (np.abs(m).max(axis=1) > 0)
        masked |= (np.abs(sim.moments.reshape(-1, 4)[:, 3]) <= FLOOR) & (np.abs(sim.moments.reshape(-1, 4)).max(axis=1) > 0)
        for name, arr in (("moments01_norm", sim.norm), ("moments01_avgA", sim.avg_A)):
            want, got = get("density%d/%s" % (k, name)).reshape(-1, 4)[~masked], arr.reshape(-1, 4)[~masked]
            assert np.array_equal(np.isnan(got), np.isnan(want)), (k, name)
            ok = ~np.isnan(want)
            tol = 4 * np.spacing(np.abs(want[ok])) + 1e-30
            if name.endswith("avgA"):   # 0.01 N + 0.99 avg: N's rounding enters scaled, earlier frames' stay
                nrm = get("density%d/moments01_norm" % k).reshape(-1, 4)[~masked]
                tol = 8 * k * np.spacing(np.maximum(np.abs(0.01 * nrm[ok]), np.abs(want[ok]))) + 1e-30
            assert np.all(np.abs(got[ok] - want[ok]) <= tol), (k, name, np.abs(got[ok] - want[ok]).max())
    assert masked.mean() < 0.1      # the isolated sprites of the probe are mostly rim


def test_the_ideal_sprite_differs_exactly_where_the_rasteriser_snaps(scene):
    """deposit() (window coordinates of infinite precision, whole-point clipping) against the rasteriser: a particle's
    footprint moves by one cell when its window coordinate lies within 1/32 pixel of a pixel edge — about 1/16 of the
    particles per axis at 4 sub-pixel bits — and points outside the unit square are cropped instead of dropped."""
    meta, get = scene
    nr, nz = meta["spec"]["nr"], meta["spec"]["nz"]
    sim = replay(meta, get, raster_bits=BITS)
    moved = total = 0
    for k in range(1, meta["frames"] + 1):
        sim.pos_A[:] = get("step%d/position_A" % k)
        ci, cj = sim.raster_cells()
        ideal = sim.deposit_cells()      # ic + (nr+1) jc, -1 when clipped
        seen = (ideal >= 0) & (ci > -(2 ** 31))
        ii, jj = ideal[seen] % (nr + 1), ideal[seen] // (nr + 1)
        d_i, d_j = ii - ci[seen], jj - cj[seen]
        assert set(np.unique(d_i)) <= {0, 1} and set(np.unique(d_j)) <= {-1, 0, 1}
        moved += int(((d_i != 0) | (d_j != 0)).sum())
        total += int(seen.sum())
    if meta["spec"]["nparticles"] ** 2 >= 100:
        assert 0.04 < moved / total < 0.25, (moved, total)


def test_probe_pins_the_snapping_rule():
    """Isolated particles at chosen sub-pixel offsets (webgl_probe): the footprint's first column is
    ceil(rint(16 x) / 16 - 5.5 - 0.5) in pixel-centre coordinates — offsets <= 1/32 snap down onto the pixel edge and lose
    a column on the right, everything else keeps the ideal footprint; rows likewise with y running downwards (offsets
    >= 31/32 gain a row at the top)."""
    meta, get = load_scene("webgl_probe")
    a = get("density1/moments01").reshape(64, 64, 4)[:, :, 3]
    st = get("init/stamp_red").reshape(11, 11)
    thresh = 1e-3 * 0.001 * st.max()
    for k in range(25):
        ic, jc = 6 + 12 * (k % 5), 6 + 12 * (k // 5)
        fr, fz = meta["offsets_r"][k], meta["offsets_z"][k]
        block = a[jc - 6: jc + 6, ic - 6: ic + 6] > thresh
        cols, rows = np.flatnonzero(block.any(axis=0)) + ic - 6, np.flatnonzero(block.any(axis=1)) + jc - 6
        centre_i, centre_j = (cols[0] + cols[-1]) // 2, (rows[0] + rows[-1]) // 2
        assert centre_i == (ic - 1 if fr <= 1 / 32 else ic), (k, fr)
        assert centre_j == (jc + 1 if fz >= 31 / 32 else jc), (k, fz)
        assert abs(a[jc - 6: jc + 6, ic - 6: ic + 6].sum() - 0.001) < 1e-9


def test_software_evaluator_against_the_real_compiler():
    """The swgl_* fixtures of round 1 (our evaluator of the shader text) and the webgl_* fixtures of the same scenes: identical
    wherever no transcendental is involved; the loop painter's cos() separates them afterwards by parts in 1e5."""
    for sw_name, gl_name in (("swgl_scene", "webgl_scene"), ("swgl_tall", "webgl_tall")):
        ms = load_json(sw_name + ".json")
        bs = load_f32gz(ms["file"])
        gs = lambda key: bs[ms["index"][key][0]: ms["index"][key][0] + ms["index"][key][1]]
        mw, gw = load_scene(gl_name)
        assert ms["spec"] == mw["spec"] and ms["position_in"] == mw["position_in"] and ms["rand0"] == mw["rand0"]
        for key in ("set/position_A", "set/velocity_A", "set/rand_A", "set/E", "set/B", "set/sink_mask", "painted/E"):
            assert same_bits(gs(key), gw(key)), key
        for k in range(1, ms["frames"] + 1):     # K3 never reads a field: the random state is identical throughout
            assert same_bits(gs("step%d/rand_A" % k), gw("step%d/rand_A" % k))
        for key in ("painted/B", "precalc/R1", "precalc/R2", "precalc/R3", "precalc/A", "step1/position_A", "step1/velocity_A"):
            a, b = gs(key), gw(key)
            assert np.nanmax(np.abs(a - b)) <= 1e-4 * np.nanmax(np.abs(b)), key


# ------------------------------------------------------------------------------------------- the demo at its own size

def demo_inputs(meta):
    """Regenerate the inputs from the rules the fixture states (input_rule, sink_rule, pdf_rule, entropy_rule)."""
    n = meta["spec"]["nparticles"] ** 2
    a, c = np.uint32(1103515245), np.uint32(12345)
    with np.errstate(over="ignore"):
        apow = np.multiply.accumulate(np.full(6 * n, a, dtype=np.uint32), dtype=np.uint32)
        geo = np.add.accumulate(np.concatenate([[np.uint32(1)], apow[:-1]]), dtype=np.uint32)
        x = apow * np.uint32(meta["input_seed"]) + c * geo
    u = ((x >> np.uint32(8)).astype(np.float64) * 2.0 ** -24).astype(np.float32).astype(np.float64).reshape(n, 6)
    pos = np.stack([0.2 * (u[:, 0] - 0.5), 0.2 * (u[:, 1] - 0.5), 0.2 * (u[:, 2] - 0.5) + 1], axis=1).astype(np.float32)
    vel = (0.002 * (u[:, 3:6] - 0.5)).astype(np.float32)
    nr, nz = meta["spec"]["nr"], meta["spec"]["nz"]
    sink = np.ones((nr, nz)); sink[nr - 1, :] = 0; sink[1:nr - 1, 0] = 0; sink[1:nr - 1, nz - 1] = 0
    pdf = np.zeros((nr, nz)); pdf[:50, 350:450] = 1
    a, c = np.uint32(1664525), np.uint32(1013904223)
    cnt = 4 * 1024 * 1024 + 4 * n
    with np.errstate(over="ignore"):
        apow = np.multiply.accumulate(np.full(cnt, a, dtype=np.uint32), dtype=np.uint32)
        geo = np.add.accumulate(np.concatenate([[np.uint32(1)], apow[:-1]]), dtype=np.uint32)
        x = apow * np.uint32(meta["entropy_lcg_seed"]) + c * geo
    entropy = (x[:4 * 1024 * 1024].astype(np.float64) / float(0xFFFFFFFF)).astype(np.float32)
    rand0 = (x[4 * 1024 * 1024:].astype(np.float64) / 4294967296.0).astype(np.float32)
    return pos.astype(np.float64), vel.astype(np.float64), sink, pdf, entropy, rand0


def window_put(texture, nr, nz, win, values):
    i0, i1, j0, j1 = win
    texture.reshape(nz, nr, 4)[j0:j1, i0:i1] = values.reshape(j1 - j0, i1 - i0, 4)


def test_demo_scene_at_its_own_size():
    """fusionsim.js's scene — 400 x 800 cells, 160 000 protons, two current loops — three frames under WebGL.  The fixture
    holds SHA-256 digests of every texture, every 61st particle and the touched windows.  Upload: digests equal.  Loop
    painter and precalc(): tolerance (cos).  Push from the fixture's coefficient window: digests of all 160 000 positions,
    velocities and random states equal after every frame.  Deposit: the touched window bit for bit above FLOOR."""
    meta, get = load_scene("webgl_demo")
    nr, nz, n = meta["spec"]["nr"], meta["spec"]["nz"], meta["spec"]["nparticles"] ** 2
    pos, vel, sink, pdf, entropy, rand0 = demo_inputs(meta)
    assert sha(rand0) == meta["rand0_sha256"]
    sim = OracleSim(meta["spec"], np.float32, raster_bits=BITS)
    sim.set_random_state(entropy=entropy, rand=rand0)
    sim.set(position=pos, velocity=vel, sink_mask=sink, source_pdf=pdf)
    for key, arr in (("position_A", sim.pos_A), ("velocity_A", sim.vel_A), ("rand_A", sim.rand_A), ("E", sim.E), ("B", sim.B),
                     ("sink_mask", sim.sink)):
        assert sha(arr) == meta["sha256"]["set/" + key], key
    assert sha(sim.inv_cdf.reshape(-1, 4)[:, :2]) == meta["sha256"]["set/inv_cdf_xy"]
    for call in meta["painters"]:
        getattr(sim, PAINT[call[0]])(*call[1:])
    want, got = get("painted/B@rows").reshape(-1, nr, 4), sim.B.reshape(nz, nr, 4)[::53]
    err = np.abs(got - want)[..., :3].max(axis=2) / np.abs(want).max()
    # the wire itself crosses row 0 (z = 0, r = 0.8): 2.5 % there, 1e-5 at the 99th percentile
    assert np.percentile(err, 99) <= 1e-4 and np.median(err) <= 1e-6 and err.max() <= 0.05
    sim.precalc()
    win = meta["coefficient_window"]
    for name in ("R1", "R2", "R3", "A"):
        arr = getattr(sim, name)
        w = get("precalc/%s@window" % name)
        mine = arr.reshape(nz, nr, 4)[win[2]:win[3], win[0]:win[1]]
        assert np.abs(mine.ravel() - w).max() <= 2e-6 * max(1.0, np.abs(w).max()), name     # where the particles are: far from the wires
        window_put(arr, nr, nz, win, w)
    for k in range(1, meta["frames"] + 1):
        sim.step()
        for key, arr in (("position_A", sim.pos_A), ("velocity_A", sim.vel_A), ("rand_A", sim.rand_A)):
            assert same_bits(arr.reshape(-1, 4)[::meta["particle_stride"]].ravel(), get("step%d/%s@stride" % (k, key))), (k, key)
            assert sha(arr) == meta["sha256"]["step%d/%s" % (k, key)], (k, key)
        sim.density()
        i0, i1, j0, j1 = meta["windows"]["density%d/moments01" % k]
        want = get("density%d/moments01@window" % k).reshape(j1 - j0, i1 - i0, 4)
        got = sim.moments.reshape(nz, nr, 4)
        outside = got.copy()
        outside[j0:j1, i0:i1] = 0
        top = np.abs(want).max()
        assert np.abs(outside).max() <= FLOOR * top
        big = np.abs(want) > FLOOR * top
        assert same_bits(got[j0:j1, i0:i1][big], want[big]), k
        assert np.abs(got[j0:j1, i0:i1] - want).max() <= FLOOR * top
