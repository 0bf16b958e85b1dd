"""Multi-GPU host logic for the reference-parity mode.

The reference never feeds the deposit back into the fields (empic.js:1471-1505 only
draws it), so particles are independent units: each rank owns a contiguous index
range of the particle population, the grid tables (E, B, sink, inverse CDF,
entropy) are replicated, and the single exchange per frame is an all-reduce (sum) of
the per-cell sums between the scatter and the stamp / normalise / EMA stage.  One
process per GPU; on GPUs the collective is RCCL through torch.distributed's "nccl"
backend, on CPU test ranks it is "gloo".  There is no other data-path collective.
"""


def shard_bounds(n_total, rank, world):
    """Contiguous [begin, end) of particle indices owned by `rank`; sizes differ by at most 1."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("rank %r outside world of %r" % (rank, world))
    base, rem = divmod(int(n_total), int(world))
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def device_tensor_view(ptr, nbytes, device, dtype="f4"):
    """Zero-copy torch view of a raw device allocation (fpic_device_buffer)."""
    import torch
    item = 4 if dtype == "f4" else 8

    class _Buf:
        __cuda_array_interface__ = {"shape": (nbytes // item,), "typestr": "<" + dtype, "data": (int(ptr), False), "version": 3}

    return torch.as_tensor(_Buf(), device=device)


class ShardedPusher:
    """One rank's pusher plus the frame's only exchange.

    `sim` offers precalc(), step(n), deposit(), densityFinish(); `sums` is a tensor that
    aliases sim's per-cell sums (a device view on GPU ranks, a CPU tensor in tests).

    overlap=True (GPU ranks): the sums are copied to a buffer of their own, the all-reduce and
    the stamp / normalise / EMA stage run on a side stream, and the next step() starts at once on
    the pusher's stream — the deposit is never fed back into the push (empic.js:1471-1505), so
    nothing on the pusher's stream waits for the exchange.  The copy is 16.8 MB at C2.
    """

    def __init__(self, sim, sums, group=None, stream=None, overlap=False):
        self.sim, self.sums, self.group, self.stream = sim, sums, group, stream
        self.overlap = bool(overlap) and stream is not None
        if self.overlap:
            import torch
            self.side = torch.cuda.Stream(device=sums.device, priority=-1)
            self.buf = torch.empty_like(sums)
            self.copied = torch.cuda.Event()
            self.finished = torch.cuda.Event()
            self.finished.record(self.side)

    def precalc(self):
        self.sim.precalc()

    def step(self, ncalls=1):
        self.sim.step(ncalls)

    def density(self):
        import torch.distributed as dist
        self.sim.deposit()
        if self.overlap:
            import torch
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(self.finished)      # the previous frame's finish still reads buf
                self.buf.copy_(self.sums)
                self.copied.record(self.stream)
            with torch.cuda.stream(self.side):
                self.side.wait_event(self.copied)
                if dist.is_initialized():
                    dist.all_reduce(self.buf, op=dist.ReduceOp.SUM, group=self.group)
                self.sim.densityFinishFrom(self.buf.data_ptr(), self.side.cuda_stream)
                self.finished.record(self.side)
            return
        if dist.is_initialized():  # also with a world of one: the same calls as the N-GPU launch
            if self.stream is not None:
                import torch
                with torch.cuda.stream(self.stream):
                    dist.all_reduce(self.sums, op=dist.ReduceOp.SUM, group=self.group)
            else:
                dist.all_reduce(self.sums, op=dist.ReduceOp.SUM, group=self.group)
        self.sim.densityFinish()

    def sync(self):
        if self.overlap:
            self.side.synchronize()
        self.sim.sync()
