"""The library's RCCL TRANSPORT (one handle per rank, fpic_comm_init + fpic_domain_init, every exchange issued by
fpic_precalc / fpic_step as grouped ncclSend/ncclRecv, ncclAllGather, ncclAllReduce) with more than one rank — on one
GPU.  The real RCCL refuses two ranks on one device, so the ranks are THREADS of one process and the ten RCCL entry
points are served by an in-process stand-in (tests/fake_rccl/fake_rccl.cpp, bound through FPIC_RCCL_LIBRARY) that keeps
RCCL's matching rule (k-th send of a pair meets its k-th receive, groups take effect at ncclGroupEnd).  What this
checks is the library's side: message lists, sizes, order, in-place halos, world of two where both neighbours are the
same rank.  Results must be bit-identical to one handle's (and to the in-process group transport's)."""
import json
import os
import subprocess
import sys

import pytest

from helpers import ROOT

pytestmark = pytest.mark.gpu

DRIVER = r'''
import json, os, sys, threading
sys.path.insert(0, os.path.join(sys.argv[1], "fusion-sim_amd"))
sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import fusionpic as fp
import decomp_scene as ds
sc = ds.build(fp, json.loads(sys.argv[2]))
world = sc["world"]
ref, ref_f = ds.run_one(fp, sc)
uid = fp.commUniqueId()
out, err = [None] * world, [None] * world
def rank_main(r):
    try:
        out[r] = ds.run_rank(fp, sc, r, uid)
    except Exception as e:  # a failed rank leaves the others waiting: the stand-in's patience ends them
        err[r] = repr(e)
threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
for t in threads: t.start()
for t in threads: t.join()
if any(err):
    print(json.dumps({"error": err})); sys.exit(0)
print(json.dumps(ds.compare(fp, sc, ref, ref_f, out)))
'''

_built = {}


def fake_lib(kind="threads"):
    """the stand-ins are HIP sources (a delay kernel, a reduction): built once per session by their Makefile (build() of
    __graft_entry__ has usually done it already and the .so travelled with the tree)"""
    if not _built:
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "fake_rccl")])
        _built["threads"] = os.path.join(ROOT, "tests", "fake_rccl", "libfakerccl.so")
        _built["procs"] = os.path.join(ROOT, "tests", "fake_rccl", "libfakerccl_shm.so")
    return _built[kind]


def build_fake(tmp_path=None):
    return fake_lib("threads")


# how the stand-in completes an operation (tests/fake_rccl/fake_rccl.cpp): stream-ordered like the real library (the
# default), the same with 1.5 ms spinning kernels in front of every arrival and behind every operation, or rounds 2-4's
# synchronous form
TRANSPORTS = {"stream": {}, "delayed": {"FAKE_RCCL_DELAY_US": "1500"}, "sync": {"FAKE_RCCL_MODE": "sync"}}


def run_case(tmp_path, transport="stream", fault=0, **case):
    env = dict(os.environ, FPIC_RCCL_LIBRARY=fake_lib("threads"), **TRANSPORTS[transport])
    if fault:
        env["FPIC_TEST_FAULT"] = str(fault)
    raw = subprocess.check_output([sys.executable, "-c", DRIVER, ROOT, json.dumps(case)], env=env, timeout=300)
    res = json.loads(raw.decode().strip().splitlines()[-1])
    assert "error" not in res, res
    return res


@pytest.mark.parametrize("precision", ["fp32", "fp64"])
@pytest.mark.parametrize("world,shape,ghost,every", [(2, (16, 12, 16), 2, 2), (3, (12, 16, 18), 1, 1), (4, (20, 12, 32), 3, 3), (2, (20, 24, 64), 3, 4)])
def test_rccl_transport_electrostatic_replicated_solve(tmp_path, precision, world, shape, ghost, every):
    res = run_case(tmp_path, world=world, shape=shape, ghost=ghost, every=every, em=False, distributed_solve=False, precision=precision, n=20000, seed=world)
    assert res["ids_ok"] and res["pos_same"] and res["vel_same"] and all(res["fields"].values()), res
    assert res["migrated"] > 0 and res["lost"] == 0


@pytest.mark.parametrize("world,shape,ghost", [(2, (16, 12, 16), 2), (4, (18, 16, 24), 2), (2, (32, 16, 64), 3), (4, (16, 32, 64), 2)])
def test_rccl_transport_slab_decomposed_solve(tmp_path, world, shape, ghost):
    res = run_case(tmp_path, world=world, shape=shape, ghost=ghost, every=2, em=False, distributed_solve=True, precision="fp32", n=20000, seed=7)
    # another summation order in the solve: particles agree to rounding, so a few 14-bit weights of the integer charge grid
    # may differ by a unit; its total is exact regardless
    assert res["ids_ok"] and res["pos_err"] <= 1e-4 and res["charge_total_same"] and res["charge_max_rel_diff"] <= 1e-3, res
    assert res["migrated"] > 0 and res["lost"] == 0
    if all(n & (n - 1) == 0 for n in shape):
        # power-of-two grid: the library's own transforms, slab-only arrays, the potential's ghost planes travelling beside
        # the inner gradient — and every number the one handle's
        assert res["pos_same"] and res["vel_same"] and all(res["fields"].values()), res


@pytest.mark.parametrize("world,shape,ghost", [(2, (16, 16, 16), 2), (4, (16, 32, 64), 2), (2, (32, 16, 64), 3)])
def test_rccl_transport_interface_solve(tmp_path, world, shape, ghost):
    """distributed_solve = 2 over the (stand-in) RCCL transport: ONE all-gather of two planes per rank instead of the two
    transpositions; agreement with one handle to the solve's rounding (it is not the same arithmetic)"""
    res = run_case(tmp_path, world=world, shape=shape, ghost=ghost, every=2, em=False, distributed_solve=2, precision="fp32", n=20000, seed=7)
    assert res["ids_ok"] and res["pos_err"] <= 1e-4 and res["charge_total_same"] and res["charge_max_rel_diff"] <= 1e-3, res
    assert res["migrated"] > 0 and res["lost"] == 0


@pytest.mark.parametrize("precision", ["fp32", "fp64"])
@pytest.mark.parametrize("world,shape,ghost,every", [(2, (16, 16, 64), 2, 4), (4, (16, 16, 64), 1, 2)])
def test_rccl_transport_full_em_from_a_decomposed_precalc(tmp_path, precision, world, shape, ghost, every):
    """distributed_solve on full-EM ranks over the (stand-in) RCCL transport: the initial field from the decomposed solve,
    slab-only arrays, then the cycle's exchanges — everything bit-identical to one handle started the same way"""
    res = run_case(tmp_path, world=world, shape=shape, ghost=ghost, every=every, em=True, distributed_solve=True, precalc=True, precision=precision, n=15000, seed=13)
    assert res["ids_ok"] and res["pos_same"] and res["vel_same"] and all(res["fields"].values()), res
    assert res["migrated"] > 0 and res["lost"] == 0


@pytest.mark.parametrize("precision", ["fp32", "fp64"])
@pytest.mark.parametrize("world,shape,ghost,every", [(2, (12, 10, 16), 2, 2), (3, (10, 12, 30), 3, 4), (2, (12, 16, 64), 3, 4)])
def test_rccl_transport_full_em(tmp_path, precision, world, shape, ghost, every):
    res = run_case(tmp_path, world=world, shape=shape, ghost=ghost, every=every, em=True, distributed_solve=False, precision=precision, n=15000, seed=11)
    assert res["ids_ok"] and res["pos_same"] and res["vel_same"] and all(res["fields"].values()), res
    assert res["migrated"] > 0 and res["lost"] == 0


@pytest.mark.parametrize("em", [False, True])
def test_rccl_transport_with_a_rank_that_empties(tmp_path, em):
    """Every particle starts in slab 0 and streams upwards: rank 0 loses its whole population to a migration that rides on
    the fused re-binning and then keeps stepping empty.  Its decision to migrate must stay the other ranks' decision (one
    rank alone in the count exchange while its neighbours post ghost planes would never return)."""
    if em:   # (0.26 cells per sub-step at 0.9 c: 40 sub-steps carry every particle out of the 10 planes of slab 0)
        res = run_case(tmp_path, world=3, shape=(10, 12, 30), ghost=3, every=4, em=True, distributed_solve=False, precision="fp32", n=6000, seed=5, emptying=True, frames=20)
    else:
        res = run_case(tmp_path, world=3, shape=(12, 16, 18), ghost=1, every=1, em=False, distributed_solve=False, precision="fp32", n=6000, seed=5, emptying=True, frames=7)
    assert res["lost"] == 0 and res["held"][0] == 0, (res["lost"], res["held"])
    assert res["ids_ok"] and res["pos_same"] and res["vel_same"] and all(res["fields"].values()), res


# ---- the transport under real concurrency (round 5).  The cases above run over the STREAM-ORDERED stand-in: a call only
# enqueues, and nothing but the caller's own events orders its streams.  The same scenes with spinning kernels in front of
# every arrival and behind every operation (every window a missing dependency leaves is 1.5 ms wide), and — the negative
# controls — with one of the library's two cross-stream waits REMOVED (FPIC_TEST_FAULT): the run must then differ from one
# handle's, or the cases above prove nothing about those waits.  Scenes with >= 24 planes per slab take the two-part push
# (part 1 -> ghost exchange on the communicator's stream beside the interior push -> join), the decomposed solves the
# potential's planes beside the inner gradient.
SPLIT_SCENES = {
    "electrostatic, replicated solve": dict(world=2, shape=(20, 24, 64), ghost=3, every=4, em=False, distributed_solve=False, precision="fp32", n=20000, seed=2),
    "electrostatic, transposed spectrum": dict(world=2, shape=(32, 16, 64), ghost=3, every=2, em=False, distributed_solve=True, precision="fp32", n=20000, seed=7),
    "electrostatic, interface solve, 4 ranks": dict(world=4, shape=(16, 32, 128), ghost=2, every=2, em=False, distributed_solve=2, precision="fp32", n=20000, seed=7),
    "full EM": dict(world=2, shape=(12, 16, 64), ghost=3, every=4, em=True, distributed_solve=False, precision="fp64", n=15000, seed=11),
    "full EM from a decomposed precalc": dict(world=2, shape=(16, 16, 64), ghost=2, every=4, em=True, distributed_solve=True, precalc=True, precision="fp32", n=15000, seed=13),
}


# eight ranks — the world of the driver's scaling run — fit one process as threads (as processes a GPU box allows four)
EIGHT_RANKS = {
    "electrostatic, interface solve": dict(world=8, shape=(16, 16, 128), ghost=2, every=2, em=False, distributed_solve=2, precision="fp32", n=24000, seed=3),
    "electrostatic, transposed spectrum": dict(world=8, shape=(16, 16, 128), ghost=2, every=2, em=False, distributed_solve=True, precision="fp32", n=24000, seed=3),
    "full EM from a decomposed precalc": dict(world=8, shape=(16, 16, 128), ghost=2, every=4, em=True, distributed_solve=True, precalc=True, precision="fp64", n=16000, seed=4),
}


def identical(res, scene):
    if scene["distributed_solve"] == 2:   # (another arithmetic in the solve: agreement to its rounding, the integer charge exact in total)
        return res["ids_ok"] and res["pos_err"] <= 1e-4 and res["charge_total_same"] and res["charge_max_rel_diff"] <= 1e-3
    return res["ids_ok"] and res["pos_same"] and res["vel_same"] and all(res["fields"].values())


@pytest.mark.parametrize("name", sorted(SPLIT_SCENES))
def test_rccl_transport_with_late_arrivals_and_late_producers(tmp_path, name):
    res = run_case(tmp_path, transport="delayed", **SPLIT_SCENES[name])
    assert identical(res, SPLIT_SCENES[name]), res
    assert res["migrated"] > 0 and res["lost"] == 0


@pytest.mark.parametrize("transport", ["stream", "delayed"])
@pytest.mark.parametrize("name", sorted(EIGHT_RANKS))
def test_rccl_transport_with_eight_ranks(tmp_path, name, transport):
    res = run_case(tmp_path, transport=transport, **EIGHT_RANKS[name])
    assert identical(res, EIGHT_RANKS[name]), res
    assert res["migrated"] > 0 and res["lost"] == 0


@pytest.mark.parametrize("name", ["electrostatic, replicated solve", "full EM"])
def test_rccl_transport_synchronous_stand_in_still_agrees(tmp_path, name):
    res = run_case(tmp_path, transport="sync", **SPLIT_SCENES[name])
    assert identical(res, SPLIT_SCENES[name]), res


@pytest.mark.parametrize("fault", [1, 2])
@pytest.mark.parametrize("name", ["electrostatic, replicated solve", "electrostatic, transposed spectrum", "full EM"])
def test_a_missing_stream_dependency_shows_as_wrong_bits(tmp_path, name, fault):
    """NEGATIVE CONTROL.  fault 1: the communicator's stream does not wait for the part-1 push (comm_fork's wait removed) —
    the ghost planes leave before they are complete; fault 2: the handle's stream does not wait for the exchange (comm_join's
    wait removed) — the received planes are added before they have arrived.  Under the stream-ordered stand-in with delays
    both MUST change the result; under the synchronous stand-in of rounds 2-4 neither could."""
    scene = SPLIT_SCENES[name]
    res = run_case(tmp_path, transport="delayed", fault=fault, **scene)
    assert not identical(res, scene), ("the stand-in did not notice the missing wait", fault, res)


MISUSE_DRIVER = r'''
import ctypes, json, sys, threading
hip = ctypes.CDLL("libamdhip64.so")
fake = ctypes.CDLL(sys.argv[1])
what = sys.argv[2]
class Uid(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * 128)]
fake.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, Uid, ctypes.c_int]
for f in (fake.ncclSend, fake.ncclRecv):
    f.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
fake.ncclAllGather.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
fake.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
uid = Uid(); fake.ncclGetUniqueId(ctypes.byref(uid))
CHAR, U32, F32, SUM, MAX = 0, 3, 7, 0, 2
def dev(nbytes):
    p = ctypes.c_void_p(); assert hip.hipMalloc(ctypes.byref(p), nbytes) == 0; hip.hipMemset(p, 0, nbytes); return p
world = 2
rcs = [None] * world
def rank(r):
    comm = ctypes.c_void_p()
    assert fake.ncclCommInitRank(ctypes.byref(comm), world, uid, r) == 0
    a, b = dev(4096), dev(4096)
    peer = 1 - r
    out = []
    if what == "ok":
        fake.ncclGroupStart(); out.append(fake.ncclSend(a, 100, CHAR, peer, comm, None)); out.append(fake.ncclRecv(b, 100, CHAR, peer, comm, None)); out.append(fake.ncclGroupEnd())
        out.append(fake.ncclAllReduce(a, a, 4, U32, MAX, comm, None))
        out.append(fake.ncclAllGather(ctypes.c_void_p(b.value + 64 * r), b, 64, CHAR, comm, None))
    elif what == "size_mismatch":
        fake.ncclGroupStart(); fake.ncclSend(a, 100, CHAR, peer, comm, None); fake.ncclRecv(b, 100 if r == 0 else 96, CHAR, peer, comm, None); out.append(fake.ncclGroupEnd())
    elif what == "send_without_receive":
        fake.ncclGroupStart(); fake.ncclSend(a, 100, CHAR, peer, comm, None)
        if r == 0: fake.ncclRecv(b, 100, CHAR, peer, comm, None)
        out.append(fake.ncclGroupEnd())
    elif what == "receive_without_send":
        fake.ncclGroupStart(); fake.ncclRecv(b, 100, CHAR, peer, comm, None)
        if r == 0: fake.ncclSend(a, 100, CHAR, peer, comm, None)
        out.append(fake.ncclGroupEnd())
    elif what == "self_peer":
        fake.ncclGroupStart(); fake.ncclSend(a, 8, CHAR, r, comm, None); fake.ncclRecv(b, 8, CHAR, r, comm, None); out.append(fake.ncclGroupEnd())
    elif what == "allgather_overlap":
        out.append(fake.ncclAllGather(ctypes.c_void_p(b.value + 32), b, 64, CHAR, comm, None))
    elif what == "allreduce_unsupported":
        out.append(fake.ncclAllReduce(a, a, 4, U32, SUM, comm, None))
    elif what == "unbalanced_group":
        out.append(fake.ncclGroupEnd())
    elif what == "join_twice":
        c2 = ctypes.c_void_p(); out.append(fake.ncclCommInitRank(ctypes.byref(c2), world, uid, r))
    if what not in ("ok", "unbalanced_group", "join_twice"):  # sticky: the world is broken for every later call
        fake.ncclGroupStart(); fake.ncclSend(a, 8, CHAR, peer, comm, None); fake.ncclRecv(b, 8, CHAR, peer, comm, None); out.append(fake.ncclGroupEnd())
    rcs[r] = out
ts = [threading.Thread(target=rank, args=(r,)) for r in range(world)]
for t in ts: t.start()
for t in ts: t.join()
print(json.dumps(rcs))
'''


@pytest.mark.parametrize("what", ["ok", "size_mismatch", "send_without_receive", "receive_without_send", "self_peer", "allgather_overlap", "allreduce_unsupported",
                                  "unbalanced_group", "join_twice"])
@pytest.mark.parametrize("mode", ["stream", "sync"])
def test_the_stand_in_rejects_what_the_real_library_would_not_survive(tmp_path, what, mode):
    """The stand-in is only worth something if a misuse of RCCL's rules FAILS under it (DESIGN.md section 6 lists the rules
    the transport relies on): each misuse returns an error on at least one rank, leaves the world broken for every rank's
    next call, and never hangs; the correct sequences return ncclSuccess."""
    so = build_fake(tmp_path)
    raw = subprocess.check_output([sys.executable, "-c", MISUSE_DRIVER, str(so), what], timeout=120, env=dict(os.environ, FAKE_RCCL_MODE=mode))
    rcs = json.loads(raw.decode().strip().splitlines()[-1])
    if what == "ok":
        assert all(rc == 0 for r in rcs for rc in r), rcs
    elif what in ("unbalanced_group", "join_twice"):
        assert all(r[0] != 0 for r in rcs), rcs
    else:
        assert any(r[0] != 0 for r in rcs), rcs          # the misuse itself is seen ...
        assert all(r[-1] != 0 for r in rcs), rcs         # ... and nobody carries on as if nothing had happened


RZ_DRIVER = r'''
import json, os, sys, threading
import numpy as np
sys.path.insert(0, os.path.join(sys.argv[1], "fusion-sim_amd"))
sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import fusionpic as fp
from helpers import make_spec, uniform_plasma, frame_sink
world, overlap = int(sys.argv[2]), sys.argv[3] == "1"
spec = make_spec(96, 80, 2, radius=1.0, height=2.0)
n = 60000
pos, vel, entropy, rand = uniform_plasma(n, spec, seed=5, v_th=5e-3)
sink = frame_sink(96, 80)
def build(lo, hi):
    s = fp.makeCylindricalParticlePusher(spec, count=hi - lo)
    s.set(position=pos[lo:hi], velocity=vel[lo:hi], sink_mask=sink, source_pdf=sink)
    s.setRandomState(entropy, rand[lo:hi]); s.addBZ(0.02); s.precalc()
    return s
one = build(0, n)
for _ in range(3):
    one.step(); one.density()
want, wantp = one.readDensity(np.float64), one.getParticles()
uid = fp.commUniqueId()
bounds = [n * r // world for r in range(world + 1)]
out, err = [None] * world, [None] * world
def rank_main(r):
    try:
        s = build(bounds[r], bounds[r + 1])
        s.commInit(uid, r, world, overlap=overlap)
        for _ in range(3):
            s.step(); s.density()
        out[r] = (s.readDensity(np.float64), s.getParticles()["position"])
        s.destroy()
    except Exception as e:
        err[r] = repr(e)
threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
for t in threads: t.start()
for t in threads: t.join()
if any(err):
    print(json.dumps({"error": err})); sys.exit(0)
g = out[0][0].reshape(-1, 4); w = want.reshape(-1, 4)
ok = ~np.isnan(w[:, 3])
res = {"count_err": float(np.abs(g[ok, 3] - w[ok, 3]).max() / np.abs(w[ok, 3]).max()),
       "ranks_agree": bool(all(np.array_equal(np.nan_to_num(out[r][0]), np.nan_to_num(out[0][0])) for r in range(world))),
       "particles_same": bool(np.array_equal(np.concatenate([o[1] for o in out]).view(np.uint8), wantp["position"].view(np.uint8)))}
print(json.dumps(res))
'''


@pytest.mark.parametrize("world,overlap,transport", [(2, True, "stream"), (3, True, "stream"), (2, False, "stream"), (3, False, "sync"), (2, True, "delayed"), (3, True, "delayed")])
def test_rccl_transport_parity_mode_all_reduce(tmp_path, world, overlap, transport):
    """(r,z) reference-parity mode, SURVEY 8(e) row 1, with the library's communicator and `world` ranks as threads: every
    rank ends up with the density of the WHOLE population (one ncclAllReduce of the per-cell sums per frame, on the side
    stream when overlapping), equal on all ranks bit for bit and equal to one handle's up to the summation order; the
    particles are those of one handle."""
    so = build_fake(tmp_path)
    env = dict(os.environ, FPIC_RCCL_LIBRARY=str(so), **TRANSPORTS[transport])
    raw = subprocess.check_output([sys.executable, "-c", RZ_DRIVER, ROOT, str(world), "1" if overlap else "0"], env=env, timeout=300)
    res = json.loads(raw.decode().strip().splitlines()[-1])
    assert "error" not in res, res
    assert res["ranks_agree"] and res["particles_same"] and res["count_err"] <= 1e-5, res
