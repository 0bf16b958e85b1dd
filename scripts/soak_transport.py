#!/usr/bin/env python3
"""Round 5 soak of the library's RCCL transport over the stream-ordered stand-in (ranks = threads), larger and longer than the
test suite's cases: scripts/soak_transport.py [--frames 40] > gpurun_out/r5_soak.txt.  Every case must end bit-identical to one
handle (tests/decomp_scene.py does the comparing)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_fake_rccl import DRIVER, fake_lib  # noqa: E402

frames = int(sys.argv[sys.argv.index("--frames") + 1]) if "--frames" in sys.argv else 40
CASES = {
    "electrostatic, transposed spectrum, 4 ranks, 64x64x128, 2e6 particles": dict(world=4, shape=(64, 64, 128), ghost=3, every=4, em=False, distributed_solve=True, precision="fp32", n=2_000_000, seed=21),
    "electrostatic, replicated solve, 2 ranks, 64x32x64, 1e6 particles, fp64": dict(world=2, shape=(64, 32, 64), ghost=3, every=4, em=False, distributed_solve=False, precision="fp64", n=1_000_000, seed=22),
    "full EM from a decomposed precalc, 4 ranks, 32x32x128, 1e6 particles, fp64": dict(world=4, shape=(32, 32, 128), ghost=2, every=8, em=True, distributed_solve=True, precalc=True, precision="fp64", n=1_000_000, seed=23),
    "full EM, 8 ranks, 32x32x256, 2e6 particles, fp32": dict(world=8, shape=(32, 32, 256), ghost=2, every=8, em=True, distributed_solve=True, precalc=True, precision="fp32", n=2_000_000, seed=24),
}
bad = 0
for name, case in CASES.items():
    for delay in ("0", "400"):
        env = dict(os.environ, FPIC_RCCL_LIBRARY=fake_lib("threads"), FAKE_RCCL_DELAY_US=delay)
        t0 = time.time()
        raw = subprocess.run([sys.executable, "-c", DRIVER, ROOT, json.dumps(dict(case, frames=frames))], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        try:
            res = json.loads(raw.stdout.decode().strip().splitlines()[-1])
        except Exception:
            res = {"error": raw.stderr.decode()[-500:]}
        ok = "error" not in res and res["ids_ok"] and res["pos_same"] and res["vel_same"] and all(res["fields"].values()) and res["lost"] == 0
        bad += not ok
        print("%s | delays %s us | %d frames | %s | migrated %s | %.0f s" % (name, delay, frames, "bit-identical to one handle" if ok else "MISMATCH %s" % res, res.get("migrated"), time.time() - t0), flush=True)
print("ok" if not bad else "FAILED")
sys.exit(1 if bad else 0)
