#!/usr/bin/env python3
"""make_golden_webgl.py — golden fixtures from the REFERENCE ITSELF, executed by a real WebGL implementation.

TEST INFRASTRUCTURE, BUILD CONTAINER ONLY (reads /root/reference at generation time; nothing here travels to the GPU
box except the numbers it writes under tests/golden/).

    python3 oracle/make_golden_webgl.py [--ref /root/reference] [--out tests/golden] [--only NAME ...]

How: the `kaleido` Python package ships a headless Chromium whose WebGL 1 runs on ANGLE/SwiftShader (software), with
OES_texture_float, WEBGL_color_buffer_float and EXT_float_blend.  Its `plotly` scope loads the script --plotlyjs= names;
oracle/webgl_stub.js stands there, loads the reference's UNMODIFIED utilities.js / matrix_webgl.js / spindle.js / empic.js
from --ref with an AMD define() shim, and runs the reference's own factory, set(), painters, precalc(), step() and
density() — every shader compiled by the browser's GLSL compiler and run by its rasteriser — reading every frame buffer
back through the reference's own fb.readPixels (utilities.js:701-711).  Only numbers are written:

  webgl_<scene>.json + .f32.gz   same layout as the swgl_* fixtures of oracle/make_golden.js (index of [offset, length]
                                 into one float32 blob), so tests/ can hold the oracle and the HIP library to both;
                                 inputs that are exactly float32 live in the blob too (`in/...`, `inputs_in_blob`)
  webgl_sor.json + .f32.gz       matrix_webgl.makeSORIterative, the cases of swgl_sor
  webgl_demo.json + .f32.gz      fusionsim.js's 400 x 800 / 160 000-particle scene: SHA-256 of every texture of every
                                 stage + every 61st particle + the touched window of the deposit
  webgl_rand<k>.json + .f32.gz   16 small scenes with everything drawn at random (grid, cylinder, time step, species, fields,
                                 masks, painters), inputs and every texture in the blob, the 512^2 injection table by digest
  webgl_info.json                what the GL implementation says about itself

The scenes `webgl_scene` and `webgl_tall` take their inputs from tests/golden/swgl_scene.json / swgl_tall.json unchanged,
so the software evaluator of round 1 (oracle/swgl.js + glsl_eval.js) is itself checked against a real GLSL compiler.
"""
import argparse
import base64
import gzip
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def find_kaleido():
    import kaleido
    exe = os.path.join(os.path.dirname(kaleido.__file__), "executable", "kaleido")
    if not os.path.exists(exe):
        raise SystemExit("kaleido executable not found at " + exe)
    return exe


class Browser:
    """One kaleido process; one JSON request per line in, one JSON reply per line out."""

    def __init__(self, stub):
        args = [find_kaleido(), "plotly", "--plotlyjs=" + stub, "--disable-gpu", "--allow-file-access-from-files",
                "--disable-dev-shm-usage", "--no-sandbox"]
        self.proc = subprocess.Popen(args, stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        hello = json.loads(self.proc.stdout.readline())
        if hello.get("code") != 0:
            raise RuntimeError("kaleido did not start: %r" % hello)

    def call(self, job):
        req = {"format": "json", "data": {"data": [], "layout": {"job": job}}}
        self.proc.stdin.write((json.dumps(req) + "\n").encode())
        self.proc.stdin.flush()
        line = self.proc.stdout.readline()
        if not line:
            raise RuntimeError("kaleido closed its output (job %s)" % job.get("kind"))
        rep = json.loads(line)
        if rep.get("code") != 0:
            raise RuntimeError("kaleido error: %s" % rep.get("message"))
        res = rep["result"]
        if isinstance(res, str):
            res = json.loads(res)
        if not res.get("ok"):
            raise RuntimeError("page error in job %s: %s\n%s" % (job.get("kind"), res.get("error"), res.get("stack")))
        return res["reply"]

    def fetch(self, name, length, piece=1 << 20):
        out = np.empty(length, dtype="<f4")
        for off in range(0, length, piece):
            rep = self.call({"kind": "fetch", "name": name, "offset": off, "count": piece})
            got = np.frombuffer(base64.b64decode(rep["data"]), dtype="<f4")
            out[off:off + got.size] = got
        return out

    def close(self):
        try:
            self.proc.stdin.close()
            self.proc.wait(timeout=20)
        except Exception:
            self.proc.kill()


def b64f32(a):
    return base64.b64encode(np.ascontiguousarray(a, dtype="<f4").tobytes()).decode()


def lcg_u32(seed, count):
    """x <- 1103515245 x + 12345 (mod 2^32): the input generator of oracle/make_golden.js, vectorised."""
    a, c = np.uint32(1103515245), np.uint32(12345)
    with np.errstate(over="ignore"):
        apow = np.multiply.accumulate(np.full(count, a, dtype=np.uint32), dtype=np.uint32)
        geo = np.add.accumulate(np.concatenate([[np.uint32(1)], apow[:-1]]), dtype=np.uint32)
        return apow * np.uint32(seed) + c * geo


def canonical_bytes(a):
    """float32 bytes with every NaN replaced by the canonical quiet NaN (payloads are not part of the contract)."""
    a = np.ascontiguousarray(a, dtype="<f4").copy()
    a[np.isnan(a)] = np.float32(np.nan)
    u = a.view("<u4")
    u[np.isnan(a)] = 0x7FC00000
    return u.tobytes()


def sha(a):
    return hashlib.sha256(canonical_bytes(a)).hexdigest()


class Blob:
    def __init__(self):
        self.chunks, self.index, self.offset = [], {}, 0

    def put(self, key, arr):
        arr = np.ascontiguousarray(arr, dtype="<f4").ravel()
        self.index[key] = [self.offset, int(arr.size)]
        self.chunks.append(arr.tobytes())
        self.offset += arr.size
        return self.index[key]

    def write(self, path):
        with open(path, "wb") as f:
            f.write(gzip.compress(b"".join(self.chunks), compresslevel=9, mtime=0))


def write_json(path, obj):
    with open(path, "w") as f:
        json.dump(obj, f, indent=1)
        f.write("\n")


N_RANDOM = 16      # webgl_rand0 .. webgl_rand15

WHAT = ("outputs of the reference's own host code and shaders run by a real WebGL 1 implementation (kaleido's headless "
        "Chromium, ANGLE on SwiftShader), read back with the reference's fb.readPixels; generated by "
        "oracle/make_golden_webgl.py")
ENTROPY_RULE = ("u32 stream x <- 1664525*x + 1013904223 (mod 2^32) from entropy_lcg_seed, texel value = float32(x / 0xFFFFFFFF), "
                "4*1024*1024 values in order; rand0 continues the same stream: value = x / 2^32")


# ----------------------------------------------------------------------------------------------------------- scenes
def run_pic(br, ref, name, job, out_dir, inputs_meta, full=True, keep=None):
    job = dict(job, kind="pic", ref_dir=ref)
    t0 = time.time()
    rep = br.call(job)
    if rep["gl_errors"]:
        raise RuntimeError("%s: gl.getError() != 0: %r" % (name, rep["gl_errors"]))
    if rep["n_fbos"] != 20:
        raise RuntimeError("%s: %d frame buffers (expected 20)" % (name, rep["n_fbos"]))
    blob = Blob()
    arrays = {}
    for key, length in rep["index"].items():
        arrays[key] = br.fetch(key, length)
    br.call({"kind": "drop"})
    meta = {"what": WHAT, "gl": rep["gl"], "spec": job["spec"], "frames": job["frames"], "entropy_lcg_seed": job["seed"],
            "entropy_rule": ENTROPY_RULE, "painters": job.get("painters", []), "api": rep["api"],
            "layout": "RGBA float32, texel 4*(i + width*j)", "file": name + ".f32.gz"}
    blob_inputs = inputs_meta.pop("inputs", None)
    meta.update(inputs_meta)
    # the inverse-CDF table: the frame buffer the shaders read must be the host table set() built (a copy draw), and
    # only x, y carry data; store those two channels once
    fbo, host = arrays.pop("set/inv_cdf").reshape(-1, 4), arrays.pop("set/inv_cdf_tex").reshape(-1, 4)
    same = bool(np.all((fbo == host) | (np.isnan(fbo) & np.isnan(host))))
    meta["inv_cdf_fbo_equals_host_table"] = same
    meta["inv_cdf_zw_all_zero"] = bool(np.all(fbo[:, 2:] == 0))
    meta["inv_cdf_nan_count"] = int(np.isnan(fbo[:, :2]).sum())
    if not same:
        raise RuntimeError(name + ": the inverse-CDF frame buffer differs from the host table")
    arrays["set/inv_cdf_xy"] = np.ascontiguousarray(fbo[:, :2]).ravel()
    stamp = arrays["init/stamp"].reshape(-1, 4)
    meta["stamp_gba_equal_red"] = bool(np.all(stamp[:, 1:] == stamp[:, :1]))
    arrays["init/stamp_red"] = np.ascontiguousarray(stamp[:, 0])
    if blob_inputs is not None:   # inputs that are exactly float32 live in the blob, not in the JSON (and so does the random state)
        for key, a in blob_inputs.items():
            blob.put("in/" + key, a)
        blob.put("in/rand0", arrays["init/rand0"])
    if full == "compact":   # ... and the 512^2 injection table by digest
        meta["sha256"] = {"set/inv_cdf_xy": sha(arrays["set/inv_cdf_xy"])}
        meta["sha256_rule"] = "SHA-256 of the little-endian float32 bytes with every NaN replaced by 0x7FC00000"
        for key in sorted(arrays):
            if key.startswith("init/") or key == "set/inv_cdf_xy":
                continue
            blob.put(key, arrays[key])
    elif full:
        if blob_inputs is None:
            meta["rand0"] = [float(v) for v in arrays["init/rand0"]]
        for key in sorted(arrays):
            if key.startswith("init/") and key != "init/stamp_red":
                continue
            blob.put(key, arrays[key])
    else:
        keep(meta, blob, arrays)
    meta["index"] = blob.index
    blob.write(os.path.join(out_dir, name + ".f32.gz"))
    write_json(os.path.join(out_dir, name + ".json"), meta)
    print("%-14s %6.1f s  %d snapshots, %.1f MB raw" % (name, time.time() - t0, len(arrays), 4e-6 * blob.offset))
    return arrays


def scene_from_swgl(br, ref, out_dir, src, dst):
    with open(os.path.join(out_dir, src + ".json")) as f:
        m = json.load(f)
    job = {"spec": m["spec"], "seed": m["entropy_lcg_seed"], "frames": m["frames"], "painters": m["painters"],
           "position_json": m["position_in"], "velocity_json": m["velocity_in"], "E_json": m["E_in"], "B_json": m["B_in"],
           "sink_mask_json": m["sink_in"], "source_pdf_json": m["pdf_in"]}
    inputs = {k: m[k] for k in ("position_in", "velocity_in", "E_in", "B_in", "sink_in", "pdf_in")}
    inputs["inputs_from"] = src + ".json (unchanged)"
    return run_pic(br, ref, dst, job, out_dir, inputs)


def unit(seed, count):
    """float32-representable uniforms in [0,1) with 24 bits from the LCG above."""
    return ((lcg_u32(seed, count) >> np.uint32(8)).astype(np.float64) * 2.0 ** -24).astype(np.float32)


def f32_scene(spec, seed, r_max, v, E_amp, B_amp, sink, pdf, z_lo=0.05, z_hi=0.95):
    """Inputs that are exactly float32 (so that they can be stored as float32 and re-read without loss)."""
    n = spec["nparticles"] ** 2
    nr, nz = spec["nr"], spec["nz"]
    u = unit(seed, 6 * n + 7 * nr * nz).astype(np.float64)
    p, g = u[:6 * n].reshape(n, 6), u[6 * n:].reshape(nr, nz, 7)
    rr = r_max * spec["radius"] * np.sqrt(p[:, 0])
    th = 2 * np.pi * p[:, 1]
    pos = np.stack([rr * np.cos(th), rr * np.sin(th), spec["height"] * (z_lo + (z_hi - z_lo) * p[:, 2])], axis=1).astype(np.float32)
    vel = (v * (p[:, 3:6] - 0.5)).astype(np.float32)
    E = (E_amp * (g[:, :, 0:3] - 0.5)).astype(np.float32)
    B = (B_amp * (g[:, :, 3:6] - np.array([0.5, 0.5, 0.0]))).astype(np.float32)
    ii, jj = np.meshgrid(np.arange(nr), np.arange(nz), indexing="ij")
    sk = sink(ii, jj).astype(np.float32)
    pd = pdf(ii, jj, g[:, :, 6]).astype(np.float32)
    return pos, vel, E, B, sk, pd


def scene_f32(br, ref, out_dir, name, spec, seed, input_seed, frames, painters, **kw):
    pos, vel, E, B, sk, pd = f32_scene(spec, input_seed, **kw)
    job = {"spec": spec, "seed": seed, "frames": frames, "painters": painters, "position": b64f32(pos), "velocity": b64f32(vel),
           "E": b64f32(E), "B": b64f32(B), "sink_mask": b64f32(sk), "source_pdf": b64f32(pd)}
    inputs = {"inputs_in_blob": {"position": list(pos.shape), "velocity": list(vel.shape), "E": list(E.shape), "B": list(B.shape),
                                 "sink_mask": list(sk.shape), "source_pdf": list(pd.shape)},
              "inputs": {"position": pos, "velocity": vel, "E": E, "B": B, "sink_mask": sk, "source_pdf": pd}}
    return run_pic(br, ref, name, job, out_dir, inputs)


def scene_random(br, ref, out_dir, k):
    """A small scene with everything drawn at random from seed k — grid, cylinder, time step, species, fields, masks, painters:
    breadth for the constants the factory bakes into its shader text (toFixed(20) literals re-read by the GLSL compiler) and
    for the index arithmetic on grids that are neither square nor powers of two.  Inputs are stored as float32 in the blob
    (`in/...`), not as JSON."""
    rng = np.random.default_rng(1000 + k)
    nr, nz = int(rng.integers(5, 25)), int(rng.integers(5, 25))
    electron = bool(rng.integers(0, 2))
    spec = {"radius": float(np.float32(rng.uniform(0.1, 3.0))), "height": float(np.float32(rng.uniform(0.1, 3.0))), "nr": nr, "nz": nz,
            "dt": float(10.0 ** rng.uniform(-10.5, -8.5)), "nparticles": int(rng.integers(5, 17)),
            "particle_mass": 9.109e-31 if electron else 1.67e-27, "particle_charge": -1.602e-19 if electron else 1.602e-19}
    ring = int(rng.integers(0, 3))
    keep = rng.random((nr, nz)) > rng.uniform(0.0, 0.4)
    dead_rows = rng.random(nr) < 0.25
    dead_rows[0] = False                  # (an empty first row makes the reference's set() throw: quirk Q12)
    pos, vel, E, B, sk, pd = f32_scene(spec, 5000 + k, r_max=float(rng.uniform(0.7, 1.05)), v=float(10.0 ** rng.uniform(-2.5, -0.3)),
                                       E_amp=float(10.0 ** rng.uniform(3, 6.5)) * (0 if k % 4 == 3 else 1), B_amp=float(10.0 ** rng.uniform(-2, 0.3)),
                                       sink=lambda i, j: np.where(keep[i, j] & (i < nr - ring) & (j >= ring) & (j < nz - ring), 1.0, 0.0),
                                       pdf=lambda i, j, u: np.where(dead_rows[i] | (u < 0.3), 0.0, u), z_lo=-0.02 if k % 3 == 0 else 0.05,
                                       z_hi=1.03 if k % 3 == 0 else 0.95)
    pd[0, :] = np.maximum(pd[0, :], np.float32(0.125))
    menu = [["addBZ", float(np.float32(rng.normal(0, 0.3)))], ["addBTheta", float(np.float32(rng.normal(0, 0.1)))],
            ["addCurrentZ", float(np.float32(rng.normal(0, 3e4)))],
            ["addCurrentLoop", float(np.float32(rng.uniform(0.2, 0.9) * spec["radius"])), float(np.float32(rng.uniform(0, 1) * spec["height"])), float(np.float32(rng.normal(0, 1e6)))]]
    painters = [menu[i] for i in range(4) if rng.random() < 0.6]
    name = "webgl_rand%d" % k
    job = {"spec": spec, "seed": 0x5EED1000 + k, "frames": 3, "painters": painters, "position": b64f32(pos), "velocity": b64f32(vel),
           "E": b64f32(E), "B": b64f32(B), "sink_mask": b64f32(sk), "source_pdf": b64f32(pd)}
    inputs = {"inputs_in_blob": {"position": list(pos.shape), "velocity": list(vel.shape), "E": list(E.shape), "B": list(B.shape),
                                 "sink_mask": list(sk.shape), "source_pdf": list(pd.shape)},
              "inputs": {"position": pos, "velocity": vel, "E": E, "B": B, "sink_mask": sk, "source_pdf": pd}}
    return run_pic(br, ref, name, job, out_dir, inputs, full="compact")


def scene_probe(br, ref, out_dir):
    """Rasteriser probe: isolated particles at chosen sub-pixel offsets on a 64 x 64 grid of a unit cylinder, no fields, one
    frame.  What the point-sprite deposit of a real rasteriser does with a window coordinate near a pixel edge is read
    off moments01 (each particle's 11 x 11 footprint is disjoint from the others')."""
    spec = {"radius": 1.0, "height": 1.0, "nr": 64, "nz": 64, "dt": 1e-12, "nparticles": 5, "particle_mass": 1.67e-27,
            "particle_charge": 1.602e-19}
    # 25 particles on a 5 x 5 lattice of cells (6 + 12 a, 6 + 12 b): window offsets f_r, f_z in 1/64 steps around
    # the pixel edge and the pixel centre
    fr = [0.0, 1 / 64, 2 / 64, 3 / 64, 31 / 64, 32 / 64, 33 / 64, 61 / 64, 62 / 64, 63 / 64, 1 / 128, 127 / 128, 1 / 32 + 1 / 256,
          1 / 32 - 1 / 256, 0.25, 0.75, 1 / 16, 15 / 16, 3 / 32, 29 / 32, 5 / 64, 59 / 64, 0.4, 0.6, 0.9]
    fz = fr[::-1]
    pos = []
    for k in range(25):
        a, b = k % 5, k // 5
        r = (6 + 12 * a + fr[k]) / 64.0
        z = (6 + 12 * b + fz[k]) / 64.0
        pos.append([r, 0.0, z])          # on the x axis: r-hat = x exactly
    pos = np.asarray(pos, dtype=np.float32)
    vel = np.full((25, 3), 1e-6, dtype=np.float32)
    ones = np.ones((64, 64), dtype=np.float32)
    job = {"spec": spec, "seed": 0x5EED000A, "frames": 1, "painters": [], "position": b64f32(pos), "velocity": b64f32(vel),
           "sink_mask": b64f32(ones), "source_pdf": b64f32(ones)}
    # (no E, no B: set() is not given them, so the reference's field textures keep their initial zeros, alpha included)
    inputs = {"inputs_in_blob": {"position": list(pos.shape), "velocity": list(vel.shape), "sink_mask": [64, 64], "source_pdf": [64, 64]},
              "inputs": {"position": pos, "velocity": vel, "sink_mask": ones, "source_pdf": ones}, "offsets_r": fr, "offsets_z": fz}
    return run_pic(br, ref, "webgl_probe", job, out_dir, inputs)


def scene_demo(br, ref, out_dir, frames=3, stride=61):
    """fusionsim.js:72-148: 400 x 800 cells, 160 000 protons in a 0.2 m cube, sink frame, block source, two current loops."""
    spec = {"radius": 1, "height": 2, "nr": 400, "nz": 800, "dt": 2e-9, "nparticles": 400, "particle_mass": 1.67e-27,
            "particle_charge": 1.602e-19}
    n = 160000
    u = unit(20261004, 6 * n).astype(np.float64).reshape(n, 6)
    pos = np.stack([0.2 * (u[:, 0] - 0.5), 0.2 * (u[:, 1] - 0.5), 0.2 * (u[:, 2] - 0.5) + 1], axis=1).astype(np.float32)
    vel = (0.002 * (u[:, 3:6] - 0.5)).astype(np.float32)
    sink = np.ones((400, 800), dtype=np.float32)
    sink[399, :] = 0
    sink[1:399, 0] = 0
    sink[1:399, 799] = 0
    pdf = np.zeros((400, 800), dtype=np.float32)
    pdf[:50, 350:450] = 1
    painters = [["addCurrentLoop", 0.8, 2.0, -10000000], ["addCurrentLoop", 0.8, 0.0, 10000000]]
    job = {"spec": spec, "seed": 0x5EED000B, "frames": frames, "painters": painters, "position": b64f32(pos), "velocity": b64f32(vel),
           "sink_mask": b64f32(sink), "source_pdf": b64f32(pdf)}
    inputs = {"input_rule": "u = float32((x >> 8) * 2^-24) for the u32 stream x <- 1103515245*x + 12345 from input_seed, six per particle; "
                            "position = float32(0.2*(u0-0.5), 0.2*(u1-0.5), 0.2*(u2-0.5)+1) m, velocity = float32(0.002*(u3..5 - 0.5)) c (double arithmetic, one rounding)",
              "input_seed": 20261004, "sink_rule": "1 except i = nr-1 and (1 <= i < nr-1, j in {0, nz-1})",
              "pdf_rule": "1 for i < 50, 350 <= j < 450, else 0", "particle_stride": stride}

    def keep(meta, blob, arrays):
        nr, nz = 400, 800
        # every cell a particle's nearest-cell lookup visits in these frames, + 2 cells: the coefficient window a replay needs
        lo, hi = np.array([nr, nz]), np.array([0, 0])
        for key, a in arrays.items():
            if key.endswith("/position_A"):
                p = a.reshape(-1, 4)
                r = np.sqrt((p[:, 0] * p[:, 0] + p[:, 1] * p[:, 1]).astype(np.float32))
                ok = np.isfinite(r) & np.isfinite(p[:, 2])
                ci = np.clip(np.floor(r[ok] * np.float32(nr)), 0, nr - 1).astype(int)
                cj = np.clip(np.floor(p[ok, 2] * np.float32(nz)), 0, nz - 1).astype(int)
                lo = np.minimum(lo, [ci.min(), cj.min()])
                hi = np.maximum(hi, [ci.max(), cj.max()])
        i0, j0 = max(int(lo[0]) - 2, 0), max(int(lo[1]) - 2, 0)
        i1, j1 = min(int(hi[0]) + 3, nr), min(int(hi[1]) + 3, nz)
        meta["coefficient_window"] = [i0, i1, j0, j1]
        digests = {}
        for key, a in sorted(arrays.items()):
            if key.startswith("init/"):
                continue
            digests[key] = sha(a)
            leaf = key.split("/")[1]
            if leaf in ("position_A", "velocity_A", "rand_A"):
                blob.put(key + "@stride", a.reshape(-1, 4)[::stride])
            elif key.startswith("density"):
                img = a.reshape(nz, nr, 4)
                rows = np.flatnonzero(np.any(img != 0, axis=(1, 2)))
                cols = np.flatnonzero(np.any(img != 0, axis=(0, 2)))
                b0, b1 = (int(rows[0]), int(rows[-1]) + 1) if rows.size else (0, 0)
                a0, a1 = (int(cols[0]), int(cols[-1]) + 1) if cols.size else (0, 0)
                meta.setdefault("windows", {})[key] = [a0, a1, b0, b1]
                blob.put(key + "@window", img[b0:b1, a0:a1])
            elif key.startswith("precalc") or key.startswith("painted"):
                blob.put(key + "@rows", a.reshape(nz, nr, 4)[::53])     # every 53rd z row of the whole texture
                if key.startswith("precalc") or key == "painted/B":
                    blob.put(key + "@window", a.reshape(nz, nr, 4)[j0:j1, i0:i1])
        meta["sha256"] = digests
        meta["sha256_rule"] = "SHA-256 of the little-endian float32 bytes with every NaN replaced by 0x7FC00000"
        meta["rand0_sha256"] = sha(arrays["init/rand0"])
        meta["window_rule"] = "X@window = texture[j0:j1, i0:i1, :] (z rows, r columns, RGBA); density windows [i0, i1, j0, j1] under 'windows' " \
                              "are the bounding box of the non-zero texels; X@rows = every 53rd z row; X@stride = every particle_stride-th texel"

    return run_pic(br, ref, "webgl_demo", job, out_dir, inputs, full=False, keep=keep)


def scene_sor(br, ref, out_dir):
    """The cases of swgl_sor (oracle/make_golden.js section 9), inputs read from that fixture's blob."""
    with open(os.path.join(out_dir, "swgl_sor.json")) as f:
        m = json.load(f)
    with gzip.open(os.path.join(out_dir, m["file"]), "rb") as f:
        src = np.frombuffer(f.read(), dtype="<f4")
    get = lambda at: src[at[0]:at[0] + at[1]]
    blob, cases, gl = Blob(), {}, None
    for name, c in m["cases"].items():
        A, b, x0 = get(c["A"]), get(c["b"]), get(c["x0"])
        job = {"kind": "sor", "ref_dir": ref, "name": name, "n_power": c["n_power"], "relaxation": c["relaxation"],
               "A": b64f32(A), "b": b64f32(b), "x0": b64f32(x0), "calls": [k["params"] for k in c["calls"]]}
        rep = br.call(job)
        if rep["gl_errors"]:
            raise RuntimeError("sor %s: gl errors %r" % (name, rep["gl_errors"]))
        gl = rep["gl"]
        arr = {k: br.fetch(k, n) for k, n in rep["index"].items()}
        br.call({"kind": "drop"})
        out = {"n_power": c["n_power"], "relaxation": c["relaxation"], "vec_length": rep["vec_length"], "vec_height": rep["vec_height"],
               "A": blob.put(name + "/A", A), "b": blob.put(name + "/b", b), "x0": blob.put(name + "/x0", x0),
               "x_after_init": blob.put(name + "/x_after_init", arr[name + "/x_after_init"]), "calls": []}
        for ci, call in enumerate(rep["calls"]):
            tag = "%s/call%d/" % (name, ci)
            rec = dict(call)
            for nm in ("result", "x_result", "x_guess", "x_stats", "R", "C"):
                rec[nm] = blob.put(tag + nm, arr[tag + nm])
            out["calls"].append(rec)
        cases[name] = out
    blob.write(os.path.join(out_dir, "webgl_sor.f32.gz"))
    write_json(os.path.join(out_dir, "webgl_sor.json"), {"what": WHAT + "; makeSORIterative, the cases and inputs of swgl_sor.json; "
               "arrays are [offset, length] into the float32 blob; A is row-major A[col + L*row]", "gl": gl, "file": "webgl_sor.f32.gz", "cases": cases})
    print("webgl_sor      %d cases" % len(cases))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--only", nargs="*", default=None)
    args = ap.parse_args()
    ref, out = os.path.abspath(args.ref), os.path.abspath(args.out)
    want = lambda n: args.only is None or n in args.only
    br = Browser(os.path.join(HERE, "webgl_stub.js"))
    try:
        info = br.call({"kind": "probe"})
        if not info.get("have_webgl") or not (info["OES_texture_float"] and info["WEBGL_color_buffer_float"] and info["EXT_float_blend"]):
            raise SystemExit("this Chromium has no usable float WebGL: %r" % info)
        write_json(os.path.join(out, "webgl_info.json"), info)
        print("WebGL: %s / %s, subpixel bits %s, highp %s" % (info["version"], info["unmasked_renderer"], info["subpixel_bits"], info["highp_fragment"]))
        if want("webgl_scene"):
            scene_from_swgl(br, ref, out, "swgl_scene", "webgl_scene")
        if want("webgl_tall"):
            scene_from_swgl(br, ref, out, "swgl_tall", "webgl_tall")
        if want("webgl_probe"):
            scene_probe(br, ref, out)
        if want("webgl_efield"):
            # Q1: strong E with E.B != 0 everywhere, electrons, open walls except a sink ring; 48 x 64 cells, 1024 particles
            scene_f32(br, ref, out, "webgl_efield",
                      {"radius": 0.5, "height": 0.8, "nr": 48, "nz": 64, "dt": 2e-10, "nparticles": 32, "particle_mass": 9.109e-31,
                       "particle_charge": -1.602e-19}, 0x5EED000C, 777, 4,
                      [["addBZ", 0.05], ["addBTheta", 0.02], ["addCurrentZ", 1e4]],
                      r_max=0.9, v=0.1, E_amp=5e6, B_amp=0.2,
                      sink=lambda i, j: np.where((i == 47) | (j == 0) | (j == 63), 0.0, 1.0),
                      pdf=lambda i, j, u: np.where((i < 20) & (j >= 20) & (j < 44), 0.25 + u, 0.0))
        if want("webgl_nan"):
            # heavy re-injection through NaN sites of the inverse CDF (Q3): most of the volume absorbs, the source has empty
            # rows and an empty first column, fast protons; 32 x 24 cells, 576 particles, 5 frames
            scene_f32(br, ref, out, "webgl_nan",
                      {"radius": 1.0, "height": 0.75, "nr": 32, "nz": 24, "dt": 4e-9, "nparticles": 24, "particle_mass": 1.67e-27,
                       "particle_charge": 1.602e-19}, 0x5EED000D, 31337, 5,
                      [["addBZ", 0.4], ["addCurrentLoop", 0.5, 0.375, 2e6]],
                      r_max=0.98, v=0.4, E_amp=1e6, B_amp=0.5,
                      sink=lambda i, j: np.where(((i + j) % 3 == 0) | (i >= 28) | (j < 2) | (j >= 22), 0.0, 1.0),
                      pdf=lambda i, j, u: np.where((i % 4 == 2) | (j == 0) | (i > 25), 0.0, 0.1 + u))
        for k in range(N_RANDOM):
            if want("webgl_rand%d" % k):
                scene_random(br, ref, out, k)
        if want("webgl_demo"):
            scene_demo(br, ref, out)
        if want("webgl_sor"):
            scene_sor(br, ref, out)
    finally:
        br.close()
    print("fixtures written to " + out)


if __name__ == "__main__":
    sys.exit(main())
