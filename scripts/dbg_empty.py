import sys, os, json, subprocess
sys.path.insert(0, "tests")
import test_gpu_fake_rccl as t
from pathlib import Path
so = t.build_fake(Path("/tmp"))
env = dict(os.environ, FPIC_RCCL_LIBRARY=str(so))
case = dict(world=3, shape=(12, 16, 18), ghost=1, every=1, em=False, distributed_solve=False, precision="fp32", n=6000, seed=5, emptying=True, frames=7)
for extra in ({}, {"FPIC_E_FROM_PHI": "0"}):
    raw = subprocess.check_output([sys.executable, "-c", t.DRIVER, t.ROOT, json.dumps(case)], env=dict(env, **extra), timeout=300)
    print(extra, raw.decode().strip().splitlines()[-1])
