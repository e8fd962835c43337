"""The N>1 path on CPU: world_size-2 gloo processes, each owning a shard of the streams (oracle as the per-rank
compute stand-in), results all-gathered with waveform_amd.dist.allgather_bars and compared with a single-process run."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_shard_arithmetic():
    from waveform_amd.dist import shard_streams
    for total in (1, 7, 8, 65536, 65537):
        for world in (1, 2, 3, 8):
            shards = [shard_streams(total, r, world) for r in range(world)]
            assert sum(s.count for s in shards) == total
            assert shards[0].first == 0
            for a, b in zip(shards, shards[1:]):
                assert b.first == a.first + a.count
            assert max(s.count for s in shards) - min(s.count for s in shards) <= 1
    with pytest.raises(ValueError):
        shard_streams(4, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, out_dir):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    import torch
    import torch.distributed as dist
    import scenarios
    from oracle import restate
    from tools import synth
    from waveform_amd.dist import shard_streams, allgather_bars

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = scenarios.make_config(dict(fft_size=1024, stereo=1, bars=1, interp_mode=1))
    sh = shard_streams(total, rank, world)
    local = np.zeros((sh.count, 2, 26), np.float32)
    for i in range(sh.count):
        o = restate.OracleSource(cfg)
        for t in range(3):
            o.feed_and_tick(synth.block(7, sh.first + i, 1, 2, t * 800, 800)[0])
        o.render_bars()
        local[i] = o.bars()
    full = allgather_bars(torch.from_numpy(local), sh)
    # the verification bench.py's configs4 region runs after its timed gathers: every rank's copy, block by block
    from waveform_amd.dist import verify_gathered
    assert verify_gathered(full, torch.from_numpy(local), sh)
    broken = full.clone()
    if rank == 1:
        broken[0, 0, 0] += 1.0          # one rank holds one wrong value: every rank must hear about it
    assert not verify_gathered(broken, torch.from_numpy(local), sh)
    # the timing reduction bench.py does: max over ranks
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == world
    np.save(Path(out_dir) / f"rank{rank}.npy", full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [6, 5])
def test_two_rank_gloo_allgather_matches_single_process(tmp_path, total):
    import torch.multiprocessing as mp
    sys.path.insert(0, str(ROOT / "tests"))
    import scenarios
    from oracle import restate
    from tools import synth

    port = _free_port()
    mp.spawn(_worker, args=(2, port, total, str(tmp_path)), nprocs=2, join=True)
    cfg = scenarios.make_config(dict(fft_size=1024, stereo=1, bars=1, interp_mode=1))
    want = np.zeros((total, 2, 26), np.float32)
    for s in range(total):
        o = restate.OracleSource(cfg)
        for t in range(3):
            o.feed_and_tick(synth.block(7, s, 1, 2, t * 800, 800)[0])
        o.render_bars()
        want[s] = o.bars()
    for r in range(2):
        got = np.load(tmp_path / f"rank{r}.npy")
        assert got.shape == want.shape
        assert np.array_equal(got, want), f"rank {r}: gathered bars differ from the single-process result"


def test_gather_checksum_sees_permutations():
    """the verification's checksum weighs every bit pattern by its position: two streams swapped inside a rank's block, or
    a compaction that landed one stream further on, must change it (a plain sum of the bit patterns would not)"""
    import torch
    from waveform_amd.dist import bars_checksum
    g = torch.Generator().manual_seed(5)
    bars = torch.rand((96, 2, 26), generator=g) * 400.0
    ref = bars_checksum(bars)
    swapped = bars.clone()
    swapped[[3, 40]] = swapped[[40, 3]]
    rolled = torch.roll(bars, 1, dims=0)
    assert not torch.equal(bars_checksum(swapped), ref) and not torch.equal(bars_checksum(rolled), ref)
    assert torch.equal(bars_checksum(bars.clone()), ref)
