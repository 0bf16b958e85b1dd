"""The plain-JavaScript twin of the oracle (oracle/pic_oracle.js, the "JS/CPU path"
timed by bench.py) must agree with the C oracle bit for bit: two independent
restatements of the same shaders, in two languages, fp32 and fp64."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

import pic_oracle as po
from helpers import ROOT, frame_sink, make_spec, same_bits, uniform_plasma

node = shutil.which("node")
pytestmark = pytest.mark.skipif(node is None, reason="node is not installed")

DRIVER = r"""
const fs = require('fs');
const {makeOracle} = require(process.argv[2]);
const inp = JSON.parse(fs.readFileSync(process.argv[3]));
const sim = makeOracle(inp.spec, {precision: inp.precision, raster_bits: inp.raster_bits || 0});
sim.set({E: inp.E, B: inp.B, position: inp.position, velocity: inp.velocity, sink_mask: inp.sink, source_pdf: inp.pdf});
const ent = new Float32Array(fs.readFileSync(inp.entropy_file).buffer.slice(0));
sim.setRandomState({entropy: ent, rand: Float32Array.from(inp.rand)});
sim.addBZ(0.125);
sim.precalc();
for (let k = 0; k < inp.cycles; k++) { sim.step(); sim.density(); }
const out = {};
for (const k of ['pos_A','vel_A','rand_A','R1','R2','R3','A','moments','norm','avg_A','inv_cdf','B'])
    out[k] = Buffer.from(sim[k].buffer).toString('base64');
fs.writeFileSync(process.argv[4], JSON.stringify(out));
"""


@pytest.mark.parametrize("raster_bits", [0, 4])
@pytest.mark.parametrize("precision", ["fp32", "fp64"])
def test_js_twin_matches_c_oracle(tmp_path, precision, raster_bits):
    import base64
    dtype = np.float32 if precision == "fp32" else np.float64
    spec = make_spec(24, 20, 30, radius=0.35, height=0.2)
    n = 900
    rng = np.random.default_rng(17)
    E = rng.normal(0, 2e4, size=(24, 20, 3)); B = rng.normal(0, 0.4, size=(24, 20, 3))
    sink = frame_sink(24, 20)
    pdf = rng.random((24, 20)) + 0.1
    pos, vel, entropy, rand = uniform_plasma(n, spec, seed=23, v_th=0.01)
    ent_file = tmp_path / "entropy.f32"
    entropy.tofile(ent_file)
    inp = dict(spec=spec, precision=precision, E=E.tolist(), B=B.tolist(), position=pos.tolist(), velocity=vel.tolist(),
               sink=sink.tolist(), pdf=pdf.tolist(), rand=rand.ravel().tolist(), entropy_file=str(ent_file), cycles=3, raster_bits=raster_bits)
    (tmp_path / "in.json").write_text(json.dumps(inp))
    (tmp_path / "driver.js").write_text(DRIVER)
    subprocess.check_call([node, str(tmp_path / "driver.js"), os.path.join(ROOT, "oracle", "pic_oracle.js"),
                           str(tmp_path / "in.json"), str(tmp_path / "out.json")])
    out = json.loads((tmp_path / "out.json").read_text())
    js = {k: np.frombuffer(base64.b64decode(v), dtype=dtype) for k, v in out.items()}

    sim = po.OracleSim(spec, dtype=dtype, raster_bits=raster_bits)      # (4: the rasterised sprites of the WebGL fixtures)
    sim.set(E=E, B=B, position=pos, velocity=vel, sink_mask=sink, source_pdf=pdf)
    sim.set_random_state(entropy, rand)
    sim.add_bz(0.125)
    sim.precalc()
    for _ in range(3):
        sim.step(); sim.density()
    for k in ("B", "inv_cdf", "R1", "R2", "R3", "A", "pos_A", "vel_A", "rand_A", "moments", "norm", "avg_A"):
        assert same_bits(js[k], getattr(sim, k)), k
    assert int((sim.alive() == 0).sum()) >= 0


def test_js_twin_timing_cli():
    out = subprocess.check_output([node, os.path.join(ROOT, "oracle", "pic_oracle.js"), "time", "40", "64", "0.2"])
    j = json.loads(out)
    assert j["particles"] == 1600 and j["value"] > 0 and j["cycles"] >= 1
