"""Shared helpers for the test-suite: fixture loading and synthetic inputs."""
import gzip
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def load_f32gz(name):
    with gzip.open(os.path.join(GOLDEN, name), "rb") as f:
        return np.frombuffer(f.read(), dtype="<f4").copy()


def same_bits(a, b):
    """Bit-exact float comparison that treats NaN == NaN (payload ignored)."""
    a = np.asarray(a)
    b = np.asarray(b)
    return a.shape == b.shape and bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


DEMO_PROTON = dict(dt=2e-9, particle_mass=1.67e-27, particle_charge=1.602e-19)


def make_spec(nr, nz, side, radius=1.0, height=1.0, **kw):
    s = dict(radius=radius, height=height, nr=nr, nz=nz, nparticles=side)
    s.update(DEMO_PROTON)
    s.update(kw)
    return s


def frame_sink(nr, nz):
    """sink mask of fusionsim.js:94-112: 1 inside, 0 on the outer-r column and z end rows."""
    sink = np.ones((nr, nz))
    sink[nr - 1, :] = 0
    sink[1:nr - 1, 0] = 0
    sink[1:nr - 1, nz - 1] = 0
    return sink


def uniform_plasma(n, spec, seed=0x5EEDF051, v_th=1e-3, margin=0.0):
    """SURVEY 8(d) synthetic inputs: uniform in the cylinder's volume, Maxwellian
    velocities; returns physical-unit position [n,3] (m), velocity [n,3] (units of c),
    entropy [1024*1024*4] and rand [n,4] (float32)."""
    rng = np.random.Generator(np.random.Philox(seed))
    rh = np.sqrt(rng.random(n)) * (1.0 - 2 * margin) + margin
    rh = np.maximum(rh, 1e-6)
    th = 2 * np.pi * rng.random(n)
    zh = rng.random(n) * (1.0 - 2 * margin) + margin
    pos = np.stack([rh * np.cos(th) * spec["radius"], rh * np.sin(th) * spec["radius"], zh * spec["height"]], axis=1)
    vel = rng.normal(0.0, v_th, size=(n, 3))
    entropy = rng.random(1024 * 1024 * 4, dtype=np.float32)
    rand = rng.random((n, 4), dtype=np.float32)
    return pos, vel, entropy, rand


def load_webgl(name):
    """A tests/golden/webgl_* fixture (oracle/make_golden_webgl.py): (meta, get(key) -> float32 array, inputs) with inputs =
    position / velocity / E / B / sink_mask / source_pdf as set() takes them and rand0, whether the fixture keeps them as JSON
    lists or as float32 in its blob (`in/...`)."""
    meta = load_json(name + ".json")
    blob = load_f32gz(meta["file"])
    get = lambda key: blob[meta["index"][key][0]: meta["index"][key][0] + meta["index"][key][1]]
    if "inputs_in_blob" in meta:
        inputs = {k: get("in/" + k).reshape(shape).astype(np.float64) for k, shape in meta["inputs_in_blob"].items()}
        inputs["rand0"] = get("in/rand0")
        for k in ("E", "B"):                    # a scene whose set() was not given a field
            inputs.setdefault(k, None)
    elif "position_in" in meta:
        inputs = {"position": meta["position_in"], "velocity": meta["velocity_in"], "E": meta["E_in"], "B": meta["B_in"],
                  "sink_mask": meta["sink_in"], "source_pdf": meta["pdf_in"], "rand0": np.asarray(meta["rand0"], dtype=np.float32)}
    else:
        inputs = None       # (webgl_demo: regenerated from the rules the fixture states)
    return meta, get, inputs
