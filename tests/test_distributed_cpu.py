"""N > 1 host logic on CPU: two gloo ranks each own a shard of the particles, scatter
their shard, all-reduce the per-cell sums, and finish; the result must equal the
single-rank result.  The oracle stands in for the per-rank compute (this is a test of
the sharding + exchange logic in fusionpic.multi, which bench.py uses with RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import ROOT, make_spec, uniform_plasma

SPEC = make_spec(32, 24, 40)
N = 1600
CYCLES = 3


def test_shard_bounds_cover_everything():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "fusion-sim_amd"))
    from fusionpic.multi import shard_bounds
    for n, w in ((10, 3), (1600, 2), (7, 8), (0, 2), (100000001, 8)):
        spans = [shard_bounds(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [e - b for b, e in spans]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(10, 3, 3)


class OracleRank:
    """Per-rank pusher for the CPU test: oracle compute, torch tensor as the exchanged buffer.
    The stamped moments are linear in the particles, like the per-cell sums the GPU
    path exchanges, so the same all-reduce applies."""

    def __init__(self, po, begin, end, scene):
        pos, vel, entropy, rand, B, sink = scene
        self.sim = po.OracleSim(SPEC, dtype=np.float64, count=end - begin)
        self.sim.set(B=B, position=pos[begin:end], velocity=vel[begin:end], sink_mask=sink, source_pdf=sink)
        self.sim.set_random_state(entropy, rand[begin:end])
        self.sums = torch.from_numpy(self.sim.moments)  # aliases the oracle's buffer

    def precalc(self): self.sim.precalc()
    def step(self, n=1): self.sim.step(n)
    def deposit(self): self.sim.deposit()
    def densityFinish(self): self.sim.density_finish()


def make_scene():
    rng = np.random.default_rng(77)
    B = rng.normal(0, 0.2, size=(32, 24, 3)); B[..., 2] += 0.5
    sink = np.ones((32, 24)); sink[31, :] = 0
    pos, vel, entropy, rand = uniform_plasma(N, SPEC, seed=5, v_th=5e-3)
    return pos, vel, entropy, rand, B, sink


def worker(rank, world, port, out_dir):
    import sys
    for p in (os.path.join(ROOT, "fusion-sim_amd"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    import pic_oracle as po
    from fusionpic.multi import ShardedPusher, shard_bounds
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    begin, end = shard_bounds(N, rank, world)
    r = OracleRank(po, begin, end, make_scene())
    sp = ShardedPusher(r, r.sums)
    sp.precalc()
    for _ in range(CYCLES):
        sp.step(); sp.density()
    counts = torch.tensor([end - begin], dtype=torch.int64)
    dist.all_reduce(counts)
    np.save(os.path.join(out_dir, "avg_%d.npy" % rank), r.sim.avg_A)
    np.save(os.path.join(out_dir, "count_%d.npy" % rank), counts.numpy())
    dist.destroy_process_group()


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_two_gloo_ranks_equal_one(tmp_path):
    import pic_oracle as po
    mp.spawn(worker, args=(2, free_port(), str(tmp_path)), nprocs=2, join=True)
    a0, a1 = np.load(tmp_path / "avg_0.npy"), np.load(tmp_path / "avg_1.npy")
    assert np.array_equal(a0, a1), "every rank holds the same reduced grid"
    assert int(np.load(tmp_path / "count_0.npy")[0]) == N      # integer parity: shard sizes sum to N
    single = OracleRank(po, 0, N, make_scene())
    single.precalc()
    for _ in range(CYCLES):
        single.step(); single.deposit(); single.densityFinish()
    scale = np.abs(single.sim.avg_A).max()
    assert np.abs(a0 - single.sim.avg_A).max() <= 1e-12 * scale
