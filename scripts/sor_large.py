"""One-off check of the dense solver at n_power = 7 (L = 65536, 16 GiB iteration matrix):
64-bit indexing, convergence of the natural-rows mode to the known solution, products/s.
Not part of the test-suite (17 GB of host memory, ~1 minute).  Writes gpurun_out/sor_large.json."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fusion-sim_amd"))
from fusionpic import sor  # noqa: E402

n_power = int(sys.argv[1]) if len(sys.argv) > 1 else 7
L = 4 * 4 ** n_power
rng = np.random.default_rng(77)
t0 = time.time()
A = np.empty((L, L), dtype=np.float32)
x_true = rng.random(L, dtype=np.float32) - 0.5
b = np.empty(L, dtype=np.float32)
rows = 2048
for r0 in range(0, L, rows):
    blk = (rng.random((rows, L), dtype=np.float32) - 0.5) * np.float32(1.0 / L)
    blk[np.arange(rows), r0 + np.arange(rows)] = 1.0 + rng.random(rows, dtype=np.float32)
    A[r0:r0 + rows] = blk
    b[r0:r0 + rows] = (blk.astype(np.float64) @ x_true.astype(np.float64)).astype(np.float32)
print("host matrix built in %.1f s" % (time.time() - t0), flush=True)
eq = sor.makeSORIterative({"n_power": n_power}, compat=False)
t0 = time.time()
eq.set_matrix(A).set_b(b).init_vector(np.zeros(L, dtype=np.float32))
print("uploaded in %.1f s" % (time.time() - t0), flush=True)
res = eq.solve({"tolerance": 1e-6, "max_iterations": 40})
err = float(np.abs(res["result"] - x_true).max())
eq.resetStats(); eq.profile(True); eq.iterate(20); eq.sync()
st = eq.stats()
per = st["seconds_iterate"] / st["iterations"]
out = {"n_power": n_power, "vec_length": L, "matrix_bytes": st["matrix_bytes"], "solve_iterations": res["iterations"],
       "diff": res["diff"], "max_abs_error_vs_known_solution": err, "us_per_product": 1e6 * per,
       "GBs": st["matrix_bytes"] / per / 1e9}
print(json.dumps(out))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/sor_large.json", "w"))
assert err < 1e-4 and res["iterations"] < 40
