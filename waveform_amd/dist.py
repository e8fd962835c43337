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
