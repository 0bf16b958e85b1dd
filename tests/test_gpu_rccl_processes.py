"""The library's RCCL transport with ranks that are PROCESSES — each with its own HIP context, its own copy of
libfusionpic.so, one handle, one communicator, the unique id handed over by rank 0 — i.e. the process topology of a
multi-GPU launch, on one GPU.  The ten RCCL entry points are served by tests/fake_rccl/fake_rccl_shm.cpp (host shared
memory, stream-ordered copies; bound through FPIC_RCCL_LIBRARY), because the real RCCL refuses two ranks on one device.
What it adds to tests/test_gpu_fake_rccl.py (ranks = threads): per-process state (nothing is shared but the id), the
hand-over of the id, and bench.py under torch.distributed.run exactly as the driver starts it (`--bootstrap gloo`:
rehearsal only).  At most 4 ranks: a GPU box allows six processes on its card.  Results must be bit-identical to one
handle's (electrostatic with the replicated solve or the transposed spectrum, full EM), or agree to the solve's rounding
(the interface solve, distributed_solve = 2, which is another arithmetic)."""
import json
import os
import socket
import subprocess
import sys

import pytest

from helpers import ROOT
from test_gpu_fake_rccl import SPLIT_SCENES, fake_lib, identical

pytestmark = pytest.mark.gpu

RANK = r'''
import json, os, pickle, sys
sys.path.insert(0, os.path.join(sys.argv[1], "fusion-sim_amd"))
sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import fusionpic as fp
import decomp_scene as ds
sc = ds.build(fp, json.loads(sys.argv[2]))
r, uid, where = int(sys.argv[3]), bytes.fromhex(sys.argv[4]), sys.argv[5]
try:
    out = ds.run_rank(fp, sc, r, uid)
    pickle.dump(out, open(where, "wb"))
except Exception as e:
    pickle.dump({"error": repr(e)}, open(where, "wb"))
'''

COORDINATOR = r'''
import json, os, pickle, subprocess, sys
sys.path.insert(0, os.path.join(sys.argv[1], "fusion-sim_amd"))
sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import fusionpic as fp
import decomp_scene as ds
case, tmp, rank_src = json.loads(sys.argv[2]), sys.argv[3], sys.argv[4]
sc = ds.build(fp, case)
world = sc["world"]
uid = fp.commUniqueId()            # rank 0's part in a real launch: the id is made here and handed to every process
files = [os.path.join(tmp, "rank%d.pkl" % r) for r in range(world)]
procs = [subprocess.Popen([sys.executable, rank_src, sys.argv[1], sys.argv[2], str(r), uid.hex(), files[r]]) for r in range(world)]
ref, ref_f = ds.run_one(fp, sc)    # (one handle, in this process, while the ranks run)
bad = []
for r, p in enumerate(procs):
    try:
        rc = p.wait(timeout=240)
        if rc != 0:
            bad.append("rank %d exited with %d" % (r, rc))
    except subprocess.TimeoutExpired:
        p.kill(); p.wait()
        bad.append("rank %d did not finish" % r)
out = []
for r, f in enumerate(files):
    if not os.path.exists(f):
        bad.append("rank %d left no result" % r); continue
    o = pickle.load(open(f, "rb"))
    if isinstance(o, dict) and "error" in o:
        bad.append("rank %d: %s" % (r, o["error"]))
    out.append(o)
if bad:
    print(json.dumps({"error": bad})); sys.exit(0)
print(json.dumps(ds.compare(fp, sc, ref, ref_f, out)))
'''


def run_processes(tmp_path, delay_us=0, fault=0, **case):
    env = dict(os.environ, FPIC_RCCL_LIBRARY=fake_lib("procs"), FAKE_RCCL_SHM_DIR=os.environ.get("FAKE_RCCL_SHM_DIR", "/dev/shm"))
    if delay_us:
        env["FAKE_RCCL_DELAY_US"] = str(delay_us)
    if fault:
        env["FPIC_TEST_FAULT"] = str(fault)
    rank_src = tmp_path / "rank.py"
    rank_src.write_text(RANK)
    raw = subprocess.check_output([sys.executable, "-c", COORDINATOR, ROOT, json.dumps(case), str(tmp_path), str(rank_src)], env=env, timeout=420)
    return json.loads(raw.decode().strip().splitlines()[-1])


PROCESS_SCENES = dict(SPLIT_SCENES)
PROCESS_SCENES["electrostatic, interface solve, 2 ranks"] = dict(world=2, shape=(32, 16, 64), ghost=3, every=2, em=False, distributed_solve=2, precision="fp32", n=20000, seed=7)
PROCESS_SCENES["full EM, 4 ranks, fp32"] = dict(world=4, shape=(16, 16, 64), ghost=1, every=2, em=True, distributed_solve=True, precalc=True, precision="fp32", n=15000, seed=13)
PROCESS_SCENES["electrostatic, transposed spectrum, 4 ranks, fp64"] = dict(world=4, shape=(16, 32, 64), ghost=2, every=2, em=False, distributed_solve=True, precision="fp64", n=20000, seed=7)


@pytest.mark.parametrize("name", sorted(PROCESS_SCENES))
def test_ranks_as_processes(tmp_path, name):
    scene = PROCESS_SCENES[name]
    res = run_processes(tmp_path, **scene)
    assert "error" not in res, res
    assert identical(res, scene), res
    assert res["migrated"] > 0 and res["lost"] == 0


@pytest.mark.parametrize("name", ["electrostatic, interface solve, 4 ranks", "full EM"])
def test_ranks_as_processes_with_late_arrivals(tmp_path, name):
    scene = PROCESS_SCENES[name]
    res = run_processes(tmp_path, delay_us=1500, **scene)
    assert "error" not in res, res
    assert identical(res, scene), res


@pytest.mark.parametrize("fault", [1, 2])
def test_ranks_as_processes_notice_a_missing_stream_dependency(tmp_path, fault):
    """NEGATIVE CONTROL, as in test_gpu_fake_rccl.py: without comm_fork's / comm_join's wait the result must differ."""
    scene = PROCESS_SCENES["electrostatic, replicated solve"]
    res = run_processes(tmp_path, delay_us=1500, fault=fault, **scene)
    assert "error" in res or not identical(res, scene), ("the stand-in did not notice the missing wait", fault, res)


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,solver", [(2, "poisson_fft"), (4, "poisson_fft"), (2, "yee")])
def test_bench_box_workload_under_the_launcher(tmp_path, world, solver):
    """`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N --workload box` — the driver's SCALE command for
    north_star's strong-scaling workload — at a reduced size, N processes on this one GPU (`--bootstrap gloo`, the
    communicator bound to the shared-memory stand-in).  It must print ONE line, say that it is a rehearsal, report the
    world the library itself sees, lose no particle and run every exchange the full-size job runs."""
    env = dict(os.environ, FPIC_RCCL_LIBRARY=fake_lib("procs"), MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1", "--workload", "box", "--bootstrap", "gloo", "--no-cpu-baseline",
           "--c4-grid", "64", "--c4-particles", "4e6", "--c4-solver", solver, "--c4-ghost", "2"] + (["--c4-precision", "fp64"] if solver == "yee" else [])
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=420)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == world and line["comm"]["world"] == world and "REHEARSAL" in line["rehearsal"], line
    assert line["value"] > 0 and line["lost_rank0"] == 0 and line["scaling"] == "strong", line
    if solver == "poisson_fft":
        assert "interface_planes_all_gather" in line["exchange_bytes_per_rank_per_substep"], line


def test_bench_default_command_under_the_launcher(tmp_path):
    """The driver's plain `bench.py --gpus 2` (the (r,z) headline sharded over the library's communicator with the all-reduce
    on the side stream, then the strong_c4 block) as two processes on this one GPU, at a reduced size."""
    env = dict(os.environ, FPIC_RCCL_LIBRARY=fake_lib("procs"), MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--bootstrap", "gloo", "--no-cpu-baseline", "--side", "1000", "--grid", "256",
           "--c4-grid", "64", "--c4-particles", "4e6", "--c4-ghost", "2"]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=420)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["comm"]["world"] == 2 and "comm_fallback" not in line and "REHEARSAL" in line["rehearsal"], line
    assert line["strong_c4"]["value"] > 0 and line["strong_c4"]["comm"]["world"] == 2 and line["strong_c4"]["lost_rank0"] == 0, line["strong_c4"]


@pytest.mark.parametrize("mode,world", [("rz", 2), ("rz", 3), ("box", 2)])
def test_javascript_host_drives_ranks_as_processes(mode, world):
    """`north_star`: "host code stays JavaScript".  examples/multi_gpu_node.js — INTEGRATION.md 4a as a program: one NODE
    process per rank through the N-API addon, rank 0 makes the unique id (empic.commUniqueId), the parent relays it, every rank
    calls simulation.commInit (and domainInit / domainSet for the box) and runs the unchanged frame loop.  Ranks share this one
    GPU over the shared-memory stand-in.  --check compares with ONE handle: particles bit for bit; the (r,z) density equal on
    all ranks bit for bit and equal to one handle's up to the summation order."""
    import shutil
    if shutil.which("node") is None or not os.path.exists(os.path.join(ROOT, "fusion-sim_amd", "lib", "fusionpic_napi.node")):
        pytest.skip("node or the N-API addon is not on this machine")
    env = dict(os.environ, FPIC_RCCL_LIBRARY=fake_lib("procs"))
    p = subprocess.run(["node", os.path.join(ROOT, "examples", "multi_gpu_node.js"), "--ranks", str(world), "--mode", mode, "--one-device", "--check"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.stdout.strip(), p.stderr.decode()[-2000:]
    res = json.loads(p.stdout.decode().strip().splitlines()[-1])
    assert p.returncode == 0 and res["ok"], (res, p.stderr.decode()[-2000:])
    assert all(c["world"] == world for c in res["comm"]) and sorted(c["rank"] for c in res["comm"]) == list(range(world)), res
    assert res["particles_same"], res
    if mode == "rz":
        assert res["ranks_agree"] and res["nan_sites_agree"] and res["density_rel_err"] <= 1e-5, res
    else:
        assert res["lost"] == 0 and res["migrated"] > 0 and res["every_particle_once"], res
