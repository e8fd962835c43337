"""Multi-GPU layer of the spectrum path: one process per GPU, streams sharded, optional all-gather of bar heights.

The reference has no distributed anything: sources share nothing, so the batch shards
embarrassingly -- a contiguous block of streams per rank, state resident on its GPU, no collective on the
data path.  The only exchange BASELINE.json's configs[4] asks for is the *result*: every rank's bar heights
all-gathered (RCCL over xGMI when the backend is "nccl") for a combined render.  This module holds the
rank/shard arithmetic and that gather; it is backend-agnostic (gloo on CPU tensors in the tests).
"""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class Shard:
    rank: int
    world: int
    first: int   # first global stream of this rank
    count: int   # streams owned by this rank
    total: int


def shard_streams(total_streams: int, rank: int, world: int) -> Shard:
    """Contiguous, balanced split: the first (total % world) ranks own one stream more."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    base, extra = divmod(total_streams, world)
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return Shard(rank, world, first, count, total_streams)


def allgather_bars(local, shard: Shard, group=None):
    """local: tensor [shard.count, display_channels, num_bars] on this rank's device.
    Returns [shard.total, display_channels, num_bars] in global stream order on every rank.
    Equal shards use one all_gather_into_tensor (a single RCCL collective: 1.7 MB/rank at 8192 streams x 2 x 26 bars,
    latency-bound over xGMI); ragged shards are padded to the largest and trimmed."""
    import torch
    import torch.distributed as dist

    if shard.world == 1:
        return local.clone()
    per = local.shape[1:]
    base, extra = divmod(shard.total, shard.world)
    largest = base + (1 if extra else 0)
    send = local
    if shard.count != largest:
        pad = torch.zeros((largest - shard.count,) + tuple(per), dtype=local.dtype, device=local.device)
        send = torch.cat([local, pad], dim=0)
    send = send.contiguous()
    out = torch.empty((shard.world * largest,) + tuple(per), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, send, group=group)
    if extra == 0:
        return out
    parts = []
    for r in range(shard.world):
        c = base + (1 if r < extra else 0)
        parts.append(out[r * largest:r * largest + c])
    return torch.cat(parts, dim=0)


def bars_checksum(t):
    """exact, position-dependent checksum of a float32 tensor (int64, wrapping)"""
    import torch
    # position-dependent: bit pattern i weighs (i mod 2^20) + 1, summed in wrapping int64 -- values permuted inside a block (streams
    # in another order, a compaction that landed elsewhere) change it, which a plain sum of the bit patterns would not
    bits = t.contiguous().view(torch.int32).to(torch.int64).reshape(-1)
    w = (torch.arange(bits.numel(), dtype=torch.int64, device=bits.device) & 0xFFFFF) + 1
    return (bits * w).sum().reshape(1)


def verify_gathered(full, own, shard: Shard, group=None) -> bool:
    """Every rank checks the gathered copy it received: block r of `full` ([shard.total, ...] in global stream order) must
    carry the checksum rank r computed over its own bars (`own`: [shard.count, ...]); the per-rank checksums travel by
    all_gather, the verdicts by an all_reduce(MIN) -- True on every rank only if every copy is right on every rank."""
    import torch
    import torch.distributed as dist
    mine = bars_checksum(own)
    if shard.world == 1:
        return bool(torch.equal(bars_checksum(full), mine))
    sums = [torch.empty_like(mine) for _ in range(shard.world)]
    dist.all_gather(sums, mine, group=group)
    ok = True
    for r in range(shard.world):
        sh = shard_streams(shard.total, r, shard.world)
        ok = ok and bool(torch.equal(bars_checksum(full[sh.first:sh.first + sh.count]), sums[r]))
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=full.device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(flag.item())


class BarsGather:
    """The per-tick exchange of BASELINE configs[4], overlapped and without a copy: the handle's tick kernel writes the batch's
    bars straight into one of two send buffers (wf_hip_set_bars_mirrors) and goes on with tick i+1; the all-gather of tick i
    runs on a side stream that waits only for that tick (11 us of xGMI wire time per peer against ~140 us of compute per
    tick).  launch() hands the written buffer over (wf_hip_bars_mirror_ready) and makes the other one
    the ticks' target; before the next tick is issued the host checks the event behind the gather that read THAT buffer (a
    launch old by then).  world 1: the "gathered" bars ARE the send buffer -- no collective, no copy.  FFT sizes that are not
    powers of two and batches whose display comes from a kernel of its own (fft sizes beyond a CU's LDS) keep the copy behind
    the tick (wf_hip_copy_bars_device_async): the path is picked per handle, by what wf_hip_set_bars_mirrors answers.
    A context manager: the send buffers are torch tensors the tick kernels write into, so close() (or leaving the `with` block,
    or the object being dropped) takes them away from the handle -- waiting for the ticks in flight -- before torch may reuse
    the memory."""

    def __init__(self, batch, shard: Shard, group=None):
        import os
        import torch
        from . import binding
        self.batch, self.shard, self.group = batch, shard, group
        self.zero_copy = False
        shape = (shard.count, batch.display_channels, batch.num_bars)
        self.send = [torch.empty(shape, dtype=torch.float32, device="cuda") for _ in range(2)]
        self.result = [None, None]
        self.done = [None, None]
        # (WF_BARS_GATHER_PRIORITY=high: the collective on a hardware queue of its own instead of one shared with a lane of the handle --
        # for the first run on a real node to try; see the note at the gather streams of wf_hip_multi.cpp)
        self.side = torch.cuda.Stream(priority=-1) if os.environ.get("WF_BARS_GATHER_PRIORITY") == "high" else torch.cuda.Stream()
        self.i = 0
        self.newest = None
        try:
            if os.environ.get("WF_BARS_GATHER_COPY"):  # A/B aid (tools/ab_gather.py): the copy behind the tick for everybody
                raise binding.WfHipError(-2, "WF_BARS_GATHER_COPY")
            batch.set_bars_mirrors([self.send[0].data_ptr()], [self.send[1].data_ptr()])
            self.zero_copy = True
        except binding.WfHipError as e:
            if e.code != -2:  # WF_HIP_ERR_UNSUPPORTED: not a power of two, or the display comes from a kernel of its own -> the copy behind the tick
                raise

    def close(self):
        if self.zero_copy:
            self.zero_copy = False
            self.side.synchronize()
            if getattr(self.batch, "h", None):  # (the handle may be gone already: nothing writes the buffers then)
                self.batch.set_bars_mirrors([], [])  # waits for the ticks in flight

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def launch(self):
        """call after every tick (or every few: ticks between two launches rewrite the same buffer): the gather of the newest
        tick's bars is enqueued on the side stream"""
        import torch
        if self.zero_copy:
            ptr = self.batch.bars_mirror_ready(self.side.cuda_stream)  # (a buffer no tick has written is filled from the handle's own bars)
            k = 0 if ptr == self.send[0].data_ptr() else 1
        else:
            k = self.i & 1
            if self.done[k] is not None:
                self.done[k].synchronize()  # the gather that read this send buffer two launches ago (long finished)
            self.batch.copy_bars_to_device_async(self.send[k].data_ptr(), self.side.cuda_stream)
        self.i += 1
        with torch.cuda.stream(self.side):
            self.result[k] = self.send[k] if (self.zero_copy and self.shard.world == 1) else allgather_bars(self.send[k], self.shard, self.group)
            ev = torch.cuda.Event()
            ev.record(self.side)
            self.done[k] = ev
        if self.zero_copy and self.done[k ^ 1] is not None:
            # the ticks issued from here on write the other buffer: the gather that read it (enqueued a launch ago) must have run.
            # A host wait that returns at once -- a device-side wait in front of every tick cost 4 % of the tick rate
            self.done[k ^ 1].synchronize()
        self.newest = k
        return k

    def wait(self):
        """blocks until every launched gather has run; returns the newest combined bars [total, display_channels, num_bars]
        (zero-copy at world 1: the send buffer itself -- valid until the first tick after the next launch rewrites it)"""
        self.side.synchronize()
        return self.result[self.newest] if self.newest is not None else None
