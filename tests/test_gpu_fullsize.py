"""GPU tests at BASELINE.json's full batch sizes, through size-independent properties (the oracle cannot run
8192 spectra per tick in seconds) plus oracle spot checks on the first/last streams of the batch."""
import numpy as np
import pytest

import waveform_amd as wf
from helpers import assert_db_close
from oracle import restate
from tools import synth

pytestmark = pytest.mark.gpu
SEED = synth.DEFAULT_SEED


def _oracle_rows(cfg, stream_ids, ticks, hop, bars=False):
    out, outb = [], []
    for s in stream_ids:
        o = restate.OracleSource(cfg)
        for t in range(ticks):
            o.feed_and_tick(synth.block(SEED, s, 1, cfg.capture_channels, t * hop, hop)[0])
        out.append(o.decibels())
        if bars:
            o.render_bars()
            outb.append(o.bars())
    return np.stack(out), (np.stack(outb) if bars else None)


def _oracle_ticks(cfg, stream_ids, ticks, hop, keep, bars=False):
    """per stream the rows (and bars) after each of the last `keep` ticks: [stream][tick][...]"""
    rows, bar_rows = [], []
    for s in stream_ids:
        o = restate.OracleSource(cfg)
        r, br = [], []
        for t in range(ticks):
            o.feed_and_tick(synth.block(SEED, s, 1, cfg.capture_channels, t * hop, hop)[0])
            if t >= ticks - keep:
                r.append(o.decibels())
                if bars:
                    o.render_bars()
                    br.append(o.bars())
        rows.append(np.stack(r))
        if bars:
            bar_rows.append(np.stack(br))
    return np.stack(rows), (np.stack(bar_rows) if bars else None)


def _reference_ticks(cfg, stream_ids, ticks, hop, keep, bars=False):
    """the same, played through the REFERENCE ITSELF -- oracle/_ref/libwfref.so: the reference's own translation units and FFTW,
    its generic class, update() / capture_audio / tick / render_bars -- instead of the C restatement: [stream][tick][...]"""
    import scenarios
    from oracle import wfref
    if not wfref.available():
        pytest.skip("oracle/_ref/libwfref.so not built")
    rows, bar_rows = [], []
    for s in stream_ids:
        be = scenarios.RefBackend(cfg, isa="generic")
        r, br = [], []
        for t in range(ticks):
            be.push(synth.block(SEED, s, 1, cfg.capture_channels, t * hop, hop)[0], muted=False)
            be.tick(1.0 / 60.0)
            if t >= ticks - keep:
                rec = be.observe()
                r.append(rec["db"])
                if bars:
                    br.append(rec["bars"])
        rows.append(np.stack(r))
        if bars:
            bar_rows.append(np.stack(br))
    return np.stack(rows), (np.stack(bar_rows) if bars else None)


# cfg3's checked streams: the first and last 32 of the batch, the 32 around every boundary between the
# concurrent launches wf_hip_tick issues ("lanes": lane l runs streams [4096 l / lanes, 4096 (l + 1) / lanes) -- three since round 6,
# two before), and one block in the middle of each lane
def cfg3_blocks(lanes, streams=4096):
    blocks = [(0, 32), (streams - 32, 32)]
    for l in range(lanes):
        lo, hi = streams * l // lanes, streams * (l + 1) // lanes
        if l > 0:
            blocks.append((lo - 16, 32))
        blocks.append(((lo + hi) // 2 - 16, 32))
    return tuple(sorted(set(blocks)))


def test_cfg3_full_batch_spot_checks_and_determinism():
    """configs[2]: 4096 stereo streams, FFT 4096, EMA + slope.  Device-generated audio (wf_synth) for every stream; 16 warm-up
    ticks (the EMA settles) + 8 checked ticks.  Against the oracle on every checked tick: the first and last 32 streams of the
    batch, the 32 streams around every boundary between the concurrent launches the batch is ticked as (three slices since round 6)
    and a block in the middle of every launch; two identical batches must agree bit for bit."""
    cfg = wf.Config.defaults(fft_size=4096, stereo=1, slope=1.0)
    streams, warm, checked, hop = 4096, 16, 8, 800
    ticks = warm + checked
    res, blocks = [], None
    for rep in range(2):
        got = []
        with wf.SpectrumBatch(cfg, streams, ring_frames=4096 + hop * (ticks + 1)) as b:
            assert b.launches_per_tick() >= 2, "the lane boundaries this test straddles"
            blocks = cfg3_blocks(b.launches_per_tick(), streams)
            b.push_synth(SEED, 0, hop * ticks)
            for t in range(ticks):
                b.tick(delay_frames=hop * (ticks - 1 - t))
                if t >= warm:
                    got.append(np.concatenate([b.decibels(first, n) for first, n in blocks]))
            full = b.decibels()
        res.append((np.stack(got, axis=1), full))
    assert np.array_equal(res[0][1], res[1][1]), "two identical runs differ: the kernel is not deterministic"
    assert np.array_equal(res[0][0], res[1][0])
    ids = [s for first, n in blocks for s in range(first, first + n)]
    want, _ = _oracle_ticks(cfg, ids, ticks, hop, checked)
    for k in range(checked):
        assert_db_close(res[0][0][:, k], want[:, k], f"cfg3 full batch vs oracle, tick {warm + k} (first/last 32 streams, lane boundaries, mid-lane)", deep=True)
    assert np.all(np.isfinite(res[0][1]))
    # one block -- the 32 streams across the lane boundary -- against the reference itself (libwfref.so, generic class), not its restatement
    blk = [i for i, sid in enumerate(ids) if 2032 <= sid < 2064]
    ref, _ = _reference_ticks(cfg, [ids[i] for i in blk], ticks, hop, checked)
    for k in range(checked):
        assert_db_close(res[0][0][blk, k], ref[:, k], f"cfg3 full batch vs libwfref.so, tick {warm + k} (streams 2032-2063)", deep=True)


def test_cfg3_whole_batch_equals_small_handles():
    """Every one of cfg3's 4096 streams, not a sample: the whole batch after 24 ticks must equal, bit for bit, the same streams
    replayed 64 at a time through 64-stream handles (one launch, no lanes, another workgroup count) -- and a 64-stream handle is
    checked against the oracle stream by stream (all 64 of the block that straddles the large batch's lane boundary)."""
    cfg = wf.Config.defaults(fft_size=4096, stereo=1, slope=1.0)
    streams, ticks, hop, part = 4096, 24, 800, 64
    with wf.SpectrumBatch(cfg, streams, ring_frames=4096 + hop * (ticks + 1)) as b:
        b.push_synth(SEED, 0, hop * ticks)
        for t in range(ticks):
            b.tick(delay_frames=hop * (ticks - 1 - t))
        full, full_state = b.decibels(), b.tsmooth()
    checked_base = 2048 - part // 2  # streams 2016 .. 2079
    for base in list(range(0, streams, part)) + [checked_base]:
        with wf.SpectrumBatch(cfg, part, ring_frames=4096 + hop * (ticks + 1)) as s:
            assert s.launches_per_tick() == 1
            s.push_synth(SEED, 0, hop * ticks, stream_id0=base)
            for t in range(ticks):
                s.tick(delay_frames=hop * (ticks - 1 - t))
            rows, state = s.decibels(), s.tsmooth()
        assert np.array_equal(rows, full[base:base + part]), f"streams {base}..{base + part - 1}: rows of the 4096-stream batch differ from a 64-stream handle"
        assert np.array_equal(state, full_state[base:base + part]), f"streams {base}..{base + part - 1}: smoothing state differs"
        if base == checked_base:
            want, _ = _oracle_rows(cfg, range(base, base + part), ticks, hop)
            assert_db_close(rows, want, f"64-stream handle, streams {base}..{base + part - 1} vs oracle", deep=True)


@pytest.mark.parametrize("n,streams,lanes", [(65536, 264, 2), (32768, 520, 3)])
def test_lanes_on_the_one_workgroup_per_cu_kernels(n, streams, lanes):
    """fft_size 65536 (big_whole_kernel) and 32768 take two / three lanes once the batch is two rounds of one-workgroup-per-CU
    workgroups: the whole batch after a few ticks must equal, bit for bit, the same streams replayed through small one-lane handles
    (every stream, across both lane boundaries), and one small block -- the one that straddles the first boundary -- the oracle."""
    cfg = wf.Config.defaults(fft_size=n, stereo=1, slope=1.0)
    ticks, hop, part = 5, 800, 24
    with wf.SpectrumBatch(cfg, streams, ring_frames=n + hop * (ticks + 1)) as b:
        assert b.launches_per_tick() == lanes, b.kernel_name()
        b.push_synth(SEED, 0, hop * ticks)
        for t in range(ticks):
            b.tick(delay_frames=hop * (ticks - 1 - t))
        full, full_state = b.decibels(), b.tsmooth()
    boundary = streams // lanes
    checked_base = boundary - part // 2
    for base in list(range(0, streams, part)) + [checked_base]:
        cnt = min(part, streams - base)
        with wf.SpectrumBatch(cfg, cnt, ring_frames=n + hop * (ticks + 1)) as s:
            assert s.launches_per_tick() == 1
            s.push_synth(SEED, 0, hop * ticks, stream_id0=base)
            for t in range(ticks):
                s.tick(delay_frames=hop * (ticks - 1 - t))
            rows, state = s.decibels(), s.tsmooth()
        assert np.array_equal(rows, full[base:base + cnt]), f"N = {n}, streams {base}..{base + cnt - 1}: rows of the {streams}-stream batch differ from a {cnt}-stream handle"
        assert np.array_equal(state, full_state[base:base + cnt]), f"N = {n}, streams {base}..{base + cnt - 1}: smoothing state differs"
        if base == checked_base:
            want, _ = _oracle_rows(cfg, range(base, base + 4), ticks, hop)
            assert_db_close(rows[:4], want, f"N = {n}: {cnt}-stream handle, streams {base}..{base + 3} vs oracle", deep=True)


def test_cfg2_256_consecutive_frames():
    """configs[1]: stereo, FFT 2048, Hann + magnitude + dB, no smoothing, "batch = 256 frames": 256 consecutive 60 fps frames
    of one stereo stream.  Every frame against the oracle; and the same 256 frames as ONE batch -- stream i analyses the
    window i hops back (per-stream A/V-sync delay) -- must give the same bits as the frame-by-frame run."""
    cfg = wf.Config.defaults(fft_size=2048, stereo=1, tsmoothing=wf.TSMOOTH["none"])
    frames, hop = 256, 800
    audio = synth.block(SEED, 0, 1, 2, 0, hop * frames)[0]
    o = restate.OracleSource(cfg)
    seq = []
    with wf.SpectrumBatch(cfg, 1) as b:
        for t in range(frames):
            blk = audio[:, t * hop:(t + 1) * hop]
            b.push_audio(blk[None])
            b.tick()
            o.feed_and_tick(blk)
            got = b.decibels()[0]
            assert_db_close(got, o.decibels(), f"cfg2 frame {t}", deep=True)
            seq.append(got)
    seq = np.stack(seq)
    with wf.SpectrumBatch(cfg, frames, ring_frames=2048 + hop * (frames + 1)) as b:
        b.push_audio(np.broadcast_to(audio[None], (frames, 2, hop * frames)))
        b.set_stream_delay(np.array([hop * (frames - 1 - i) for i in range(frames)], np.uint32))
        b.tick()
        batch = b.decibels()
    assert np.array_equal(batch, seq), "256 frames as one batch differ from the frame-by-frame run"


def test_cfg5_per_gpu_shape_bars_only():
    """configs[4] per GPU: 8192 stereo streams, FFT 4096, EMA + slope, 26 Lanczos bars per channel, bars-only ticks
    (WF_HIP_TICK_NO_DECIBELS); 16 warm-up + 8 checked ticks, first / last 32 streams' bars against the oracle."""
    cfg = wf.Config.defaults(fft_size=4096, stereo=1, slope=1.0, bars=1, interp_mode=wf.INTERP["lanczos"])
    streams, warm, checked, hop = 8192, 16, 8, 800
    ticks = warm + checked
    ids = list(range(32)) + list(range(streams - 32, streams))
    got = []
    with wf.SpectrumBatch(cfg, streams, ring_frames=4096 + hop * (ticks + 1)) as b:
        b.push_synth(SEED, 0, hop * ticks)
        for t in range(ticks):
            b.tick(delay_frames=hop * (ticks - 1 - t), flags=wf.TICK_NO_DECIBELS)
            if t >= warm:
                got.append(np.concatenate([b.bars(0, 32), b.bars(streams - 32, 32)]))
        assert b.bars().shape == (streams, 2, 26)
    got = np.stack(got, axis=1)
    _, want = _oracle_ticks(cfg, ids, ticks, hop, checked, bars=True)
    err = np.abs(got.astype(np.float64) - want)
    assert np.all(err <= 1e-5 * np.abs(want) + 2e-3), f"cfg5 per-GPU shape bars: max err {err.max():.3e} px"


def test_cfg5_full_size_65536_streams_in_eight_shards():
    """BASELINE configs[4] at its stated size: 65536 stereo streams, FFT 4096, EMA + slope, 26 Lanczos bars per channel, sharded
    contiguously over eight devices (wf_hip_multi_*: on a box with fewer, the shards cycle over what it has -- eight shards of
    8192 streams on the one MI355X of a gpurun box: 6 GB of rings, state and rows), bars-only ticks, the all-gather of the bar
    heights after EVERY tick.  16 warm-up + 8 checked ticks; on every checked tick the first and last 32 streams OF EVERY SHARD
    against the oracle ("first + last 32 streams per GPU"), and every device index's gathered copy must
    hold the same bits, which must be the shards' own bars in global stream order."""
    cfg = wf.Config.defaults(fft_size=4096, stereo=1, slope=1.0, bars=1, interp_mode=wf.INTERP["lanczos"])
    streams, shards, warm, checked, hop = 65536, 8, 16, 8, 800
    ticks = warm + checked
    devices = [i % wf.device_count() for i in range(shards)]
    per = streams // shards
    blocks = [(i * per + off, 32) for i in range(shards) for off in (0, per - 32)]
    ids = [s for first, n in blocks for s in range(first, first + n)]
    got = []
    with wf.MultiBatch(cfg, streams, devices) as m:
        assert m.n_devices == shards and [(s[2], s[3]) for s in m.shards] == [(i * per, per) for i in range(shards)]
        assert m.transport == ("rccl" if len(set(devices)) == shards else "peer")
        for t in range(ticks):
            m.push_synth(SEED, t * hop, hop)
            m.tick(flags=wf.TICK_NO_DECIBELS)
            m.allgather_bars()
            if t >= warm:
                g0 = m.gathered(0)
                for i in range(1, shards):
                    assert np.array_equal(m.gathered(i), g0), f"tick {t}: device index {i} holds other gathered bars than index 0"
                assert np.array_equal(g0, m.bars()), f"tick {t}: the gathered bars are not the shards' bars in stream order"
                got.append(g0[ids])
        assert m.bars().shape == (streams, 2, 26)
        assert m.algorithmic_bytes_per_tick(wf.TICK_NO_DECIBELS) == streams * 2 * (4 * 4096 + 8 * 2048 + 4 * 26)
    got = np.stack(got, axis=1)
    assert np.all(np.isfinite(got))
    _, want = _oracle_ticks(cfg, ids, ticks, hop, checked, bars=True)
    err = np.abs(got.astype(np.float64) - want)
    assert np.all(err <= 1e-5 * np.abs(want) + 2e-3), f"cfg5 full size, bars of the first / last 32 streams of every shard: max err {err.max():.3e} px"
    # the first 32 streams of the LAST shard against the reference itself (libwfref.so, generic class)
    at = ids.index((shards - 1) * per)
    _, refb = _reference_ticks(cfg, ids[at:at + 32], ticks, hop, checked, bars=True)
    err = np.abs(got[at:at + 32].astype(np.float64) - refb)
    assert np.all(err <= 1e-5 * np.abs(refb) + 2e-3), f"cfg5 full size vs libwfref.so, streams {ids[at]}..{ids[at + 31]}: max err {err.max():.3e} px"


def test_bars_gather_world1_and_self_launching_bench():
    """The configs[4] exchange at world 1 runs the whole device path -- wf_hip_copy_bars_device_async onto a side stream, the
    double-buffered BarsGather under the following tick -- and must hand back the bars of the tick it was launched after;
    bench.py --gpus 8 launches itself (no torch.distributed.run wrapper) and reports the devices it measured.  Both in
    child processes: torch brings its own HIP runtime and has to be imported before libwaveform_hip.so is loaded."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "tests" / "gather_world1.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "gather ok" in r.stdout, (r.stdout[-1000:], r.stderr[-2000:])
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "8", "--steps", "4", "--warmup", "2", "--streams", "256",
                        "--no-cpu-baseline", "--no-other-configs", "--bars-allgather"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert 1 <= line["n_gpus"] <= 8
    assert line["value"] > 0 and line["roofline"]["frac"] > 0
    # the scaling point travels in a standard key (config) and again as the line's LAST key (the driver keeps the standard keys whole
    # and the last 8 KB of stdout)
    sp = line["config"]["scaling_point"]
    assert sp["n"] == line["n_gpus"] and sp["spectra_per_s"] == line["value"] and list(line)[-1] == "summary" and line["summary"]["scaling_point"] == sp
    if sp["n"] == 1:
        assert sp["efficiency_vs_n1"] == 1.0


def test_bench_multi_rank_path_with_the_collective():
    """bench.py as the driver launches it for N > 1 -- torch.distributed.run, one process per rank, the headline region, then
    BASELINE configs[4] with the all-gather of the bar heights under the next tick and the per-rank checksum verification --
    on this box's devices: with fewer devices than ranks the ranks share them and the collectives run on gloo (test aid
    WF_BENCH_SHARE_DEVICES; on a box with two devices and more this is RCCL).  The line must carry n_gpus = 2 and a verified
    configs4 object with both ranks' device times."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    import socket
    root = Path(__file__).resolve().parent.parent
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    if wf.device_count() < 2:
        env["WF_BENCH_SHARE_DEVICES"] = "1"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(root / "bench.py"), "--gpus", "2", "--steps", "30", "--warmup", "5", "--lead-in-ms", "5"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == "weak"
    assert line["ranks"]["world"] == 2 and line["ranks"]["seen_by_all_reduce"] == 2 and len(line["ranks"]["ms_per_step_per_rank"]) == 2
    c4 = line["configs4"]
    assert "error" not in c4, c4
    assert c4["verified"] is True and c4["streams_total"] == 2 * 8192 and len(c4["device_ms_per_tick"]["per_rank"]) == 2
    assert "all_gather" in c4["collective"]
    sp = line["config"]["scaling_point"]  # N = 2: the N = 1 point is rank 0 alone in the same process group
    assert sp["n"] == 2 and sp["n1_spectra_per_s"] > 0 and 0.05 < sp["efficiency_vs_n1"] < 1.5 and "rank 0 alone" in sp["n1_how"]
    assert list(line)[-1] == "summary" and "configs4 (process group, gather under the next tick)" in line["summary"]["frac_of_8TBps_by_shape"]
    assert line["roofline"]["summary"] == line["summary"]


def test_roctx_ranges_are_opt_in_and_change_nothing():
    """WF_HIP_ROCTX=1: wf_hip_tick pushes / pops a roctx range around itself, resolved at run time from the profiler's marker library
    ("Tracing"); the rows are the same bits, and without a marker library on the box the tick simply runs"""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    code = r'''
import sys, hashlib
sys.path.insert(0, %r)
import waveform_amd as wf
from tools import synth
cfg = wf.Config.defaults(fft_size=2048, stereo=1, slope=1.0, bars=1, interp_mode=wf.INTERP["lanczos"])
with wf.SpectrumBatch(cfg, 16) as b:
    for t in range(4):
        b.push_synth(synth.DEFAULT_SEED, t * 800, 800)
        b.tick()
    print("rows", hashlib.sha256(b.decibels().tobytes() + b.bars().tobytes()).hexdigest())
''' % str(root)
    import os
    outs = []
    for v in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, WF_HIP_ROCTX=v))
        assert r.returncode == 0, r.stderr[-1500:]
        outs.append([l for l in r.stdout.splitlines() if l.startswith("rows")][-1])
    assert outs[0] == outs[1]


def test_cfg3_gain_invariance_full_batch():
    """linearity of the path up to the dB stage: doubling every sample adds exactly 20*log10(2) dB
    (float scaling by 2 is exact through window, FFT, |X|, slope and the EMA)."""
    cfg = wf.Config.defaults(fft_size=4096, stereo=1, slope=1.0)
    streams, ticks, hop = 512, 4, 800
    audio = [synth.block(SEED, 0, streams, 2, t * hop, hop) for t in range(ticks)]
    outs = []
    for gain in (1.0, 2.0):
        with wf.SpectrumBatch(cfg, streams) as b:
            for t in range(ticks):
                b.push_audio(audio[t] * np.float32(gain))
                b.tick()
            outs.append(b.decibels())
    diff = outs[1].astype(np.float64) - outs[0].astype(np.float64)
    assert np.max(np.abs(diff - 20 * np.log10(2.0))) < 3e-5


def test_batch_position_independence():
    """stream i computes the same bits whether it sits in a batch of 3 or of 4096 (different workgroup, same math)"""
    cfg = wf.Config.defaults(fft_size=4096, stereo=1, slope=1.0)
    ticks, hop = 4, 800
    with wf.SpectrumBatch(cfg, 4096, ring_frames=4096 + hop * (ticks + 1)) as big:
        big.push_synth(SEED, 0, hop * ticks)
        for t in range(ticks):
            big.tick(delay_frames=hop * (ticks - 1 - t))
        full = big.decibels(first=4000, count=3)
    with wf.SpectrumBatch(cfg, 3, ring_frames=4096 + hop * (ticks + 1)) as small:
        small.push_synth(SEED, 0, hop * ticks, stream_id0=4000)
        for t in range(ticks):
            small.tick(delay_frames=hop * (ticks - 1 - t))
        part = small.decibels()
    assert np.array_equal(full, part)


def test_cfg4_full_batch_bars():
    """configs[3]: FFT 16384, gravity (TV-EMA) smoothing, 26 Lanczos bars, 1024 stereo streams.  16 warm-up ticks (the TV-EMA at
    g = 0.876 needs them to settle) + 8 checked ticks; the first and last 32 streams of the batch -- rows and bars -- against
    the oracle on every checked tick; two identical batches must agree bit for bit."""
    cfg = wf.Config.defaults(fft_size=16384, stereo=1, tsmoothing=wf.TSMOOTH["tvexponential"], bars=1, interp_mode=wf.INTERP["lanczos"])
    streams, warm, checked, hop = 1024, 16, 8, 800
    ticks = warm + checked
    ids = list(range(32)) + list(range(streams - 32, streams))
    res = []
    for rep in range(2):
        rows, bars = [], []
        with wf.SpectrumBatch(cfg, streams, ring_frames=16384 + hop * (ticks + 1)) as b:
            b.push_synth(SEED, 0, hop * ticks)
            for t in range(ticks):
                b.tick(delay_frames=hop * (ticks - 1 - t))
                if t >= warm:
                    rows.append(np.concatenate([b.decibels(0, 32), b.decibels(streams - 32, 32)]))
                    bars.append(np.concatenate([b.bars(0, 32), b.bars(streams - 32, 32)]))
            full_rows, full_bars = b.decibels(), b.bars()
        res.append((np.stack(rows, axis=1), np.stack(bars, axis=1), full_rows, full_bars))
    for a, c, what in zip(res[0], res[1], ("checked rows", "checked bars", "rows of the whole batch", "bars of the whole batch")):
        assert np.array_equal(a, c), f"cfg4: {what} differ between two identical runs"
    assert res[0][3].shape == (streams, 2, 26) and np.all(np.isfinite(res[0][3])) and np.all(np.isfinite(res[0][2]))
    want, wantb = _oracle_ticks(cfg, ids, ticks, hop, checked, bars=True)
    for k in range(checked):
        assert_db_close(res[0][0][:, k], want[:, k], f"cfg4 full batch rows vs oracle, tick {warm + k} (first/last 32 streams)", deep=True)
        err = np.abs(res[0][1][:, k].astype(np.float64) - wantb[:, k])
        assert np.all(err <= 1e-5 * np.abs(wantb[:, k]) + 2e-3), f"cfg4 bars, tick {warm + k}: max err {err.max():.3e} px"
    # the last 32 streams of the batch against the reference itself (libwfref.so, generic class), rows and bars
    ref, refb = _reference_ticks(cfg, ids[32:], ticks, hop, checked, bars=True)
    for k in range(checked):
        assert_db_close(res[0][0][32:, k], ref[:, k], f"cfg4 full batch rows vs libwfref.so, tick {warm + k} (last 32 streams)", deep=True)
        err = np.abs(res[0][1][32:, k].astype(np.float64) - refb[:, k])
        assert np.all(err <= 1e-5 * np.abs(refb[:, k]) + 2e-3), f"cfg4 bars vs libwfref.so, tick {warm + k}: max err {err.max():.3e} px"


def test_bars_only_mode_matches_full_mode():
    """WF_HIP_TICK_NO_DECIBELS (configs[4]'s batch mode): the bars are the same, the m_decibels rows are left untouched"""
    cfg = wf.Config.defaults(fft_size=4096, stereo=1, slope=1.0, bars=1, interp_mode=wf.INTERP["lanczos"])
    streams, ticks, hop = 64, 3, 800
    res = []
    for flags in (0, 1):
        with wf.SpectrumBatch(cfg, streams) as b:
            for t in range(ticks):
                b.push_synth(SEED, t * hop, hop)
                b.tick(flags=flags)
            res.append((b.bars(), b.decibels()))
    assert np.array_equal(res[0][0], res[1][0])
    assert np.all(res[1][1] == np.float32(wf.db_min()))


@pytest.mark.parametrize("layout", [
    dict(fft_size=1024, stereo=1), dict(fft_size=4096, stereo=1), dict(fft_size=2048, stereo=1, capture_channels=1),
    dict(fft_size=4096, stereo=0, capture_channels=1), dict(fft_size=2048, stereo=0), dict(fft_size=512, stereo=1),
    dict(fft_size=8192, stereo=1), dict(fft_size=800, stereo=1), dict(fft_size=8192, stereo=0, capture_channels=1),
    dict(fft_size=8000, stereo=1), dict(fft_size=1600, stereo=0), dict(fft_size=4160, stereo=1), dict(fft_size=1120, stereo=1)])
def test_bars_only_mode_keeps_the_silence_state_machine(layout):
    """WF_HIP_TICK_NO_DECIBELS through noise -> digital silence (the display decays below floor - 10, m_last_silent latches,
    reference src/source_generic.cpp:74-95) -> one live channel (the skipped one is re-dBFS'ed) -> hide/show -> noise:
    bars and m_last_silent of a bars-only batch equal the full mode's on every tick, on the geometries that keep a stereo
    pair in one workgroup, on the split ones, zero-padded, mixed radix (800, 1600, 8000), Bluestein (4160, 1120) and mono mixdown
    (which stores its row regardless)."""
    cfg = wf.Config.defaults(slope=1.0, bars=1, interp_mode=wf.INTERP["lanczos"], gravity=0.2, **layout)
    n, streams, hop = cfg.fft_size, 5, 800
    cc = int(cfg.capture_channels)
    script = [("noise", hop)] * 3 + [("silence", n + 400)] + [("silence", hop)] * 14 + [("ch0", n + 400)] + [("ch0", hop)] * 3 \
        + [("noise", hop)] * 2 + [("hide", hop)] * 2 + [("show", hop)] + [("silence", n + 400)] + [("silence", hop)] * 14 + [("noise", hop)] * 2
    res = []
    for flags in (0, 1):
        out = []
        pos = 0
        with wf.SpectrumBatch(cfg, streams) as b:
            for op, frames in script:
                a = synth.block(SEED, 0, streams, cc, pos, frames)
                pos += frames
                if op == "silence":
                    a[:] = 0.0
                elif op == "ch0" and cc > 1:
                    a[:, 1] = 0.0
                if op == "hide":
                    b.set_hidden(np.ones(streams, np.uint8))
                elif op == "show":
                    b.set_hidden(np.zeros(streams, np.uint8))
                # stream 3 stays live throughout: a batch with streams in different states
                a[3] = synth.block(SEED, 7, 1, cc, pos, frames)[0]
                b.push_audio(a)
                b.tick(flags=flags)
                out.append((b.bars(), b.last_silent()))
        res.append(out)
    went_silent = False
    for t, ((b0, s0), (b1, s1)) in enumerate(zip(*res)):
        assert np.array_equal(s0, s1), f"tick {t} ({script[t][0]}): m_last_silent {s0} vs bars-only {s1}"
        assert np.array_equal(b0, b1), f"tick {t} ({script[t][0]}): bars differ in bars-only mode"
        went_silent = went_silent or bool(s0[0])
    assert went_silent, "the script must drive stream 0 into m_last_silent"


def test_delay_frames_is_the_av_sync_window():
    """tick(delay_frames=d) analyses the window that ends d frames before the newest sample (reference :50-59)"""
    cfg = wf.Config.defaults(fft_size=2048, stereo=1, tsmoothing=wf.TSMOOTH["none"])
    hop, ticks = 800, 4
    audio = synth.block(SEED, 0, 2, 2, 0, hop * ticks)
    with wf.SpectrumBatch(cfg, 2, ring_frames=2048 + hop * (ticks + 1)) as b:
        b.push_audio(audio)
        got = []
        for t in range(ticks):
            b.tick(delay_frames=hop * (ticks - 1 - t))
            got.append(b.decibels())
    for s in range(2):
        o = restate.OracleSource(cfg)
        for t in range(ticks):
            o.feed_and_tick(audio[s][:, t * hop:(t + 1) * hop])
            assert_db_close(got[t][s], o.decibels(), f"delay tick {t} stream {s}", deep=True)


def test_errors_are_reported_not_thrown():
    with pytest.raises(wf.WfHipError) as e:
        wf.SpectrumBatch(wf.Config.defaults(fft_size=65552), 1)
    assert e.value.code == -2  # WF_HIP_ERR_UNSUPPORTED: above the reference's own maximum (65536, "enable large FFT")
    cfg = wf.Config.defaults(fft_size=1024)
    with wf.SpectrumBatch(cfg, 2) as b:
        with pytest.raises(wf.WfHipError):
            b.tick(delay_frames=10**6)
        with pytest.raises(wf.WfHipError):
            b.bars()
        with pytest.raises(wf.WfHipError):
            b.decibels(first=2, count=1)


def test_pipelined_ingest_matches_the_blocking_one():
    """wf_hip_push_audio_async (page-locked buffers, copy stream, two slots) feeds the rings exactly as wf_hip_push_audio does"""
    cfg = wf.Config.defaults(fft_size=2048, stereo=1, slope=1.0, bars=1, interp_mode=wf.INTERP["lanczos"])
    streams, ticks, hop = 96, 7, 800
    audio = [synth.block(SEED, 0, streams, 2, t * hop, hop) for t in range(ticks)]
    with wf.SpectrumBatch(cfg, streams) as ref:
        for t in range(ticks):
            ref.push_audio(audio[t])
            ref.tick()
        want_db, want_bars = ref.decibels(), ref.bars()
    pin = [wf.PinnedBuffer((streams, 2, hop)), wf.PinnedBuffer((streams, 2, hop))]
    out = [wf.PinnedBuffer((streams, 2, 26)), wf.PinnedBuffer((streams, 2, 26))]
    per_tick = []
    with wf.SpectrumBatch(cfg, streams) as ref2:
        for t in range(ticks):
            ref2.push_audio(audio[t])
            ref2.tick()
            per_tick.append(ref2.bars())
    with wf.SpectrumBatch(cfg, streams) as b:
        for t in range(ticks):
            slot = t & 1
            b.ingest_done(slot)
            pin[slot].array[...] = audio[t]
            b.push_audio_async(pin[slot], streams, hop, slot)
            b.tick()
            b.readback_done(slot)
            if t >= 2:  # what the slot received two ticks ago: the bars of tick t - 2
                assert np.array_equal(out[slot].array, per_tick[t - 2]), f"pipelined readback of tick {t - 2}"
            b.read_bars_async(out[slot], slot)
        got_db, got_bars = b.decibels(), b.bars()
    for p in pin + out:
        p.close()
    assert np.array_equal(got_db, want_db) and np.array_equal(got_bars, want_bars)


def test_two_handles_on_two_host_threads():
    """distinct handles may be used concurrently (SURVEY.md section 8(b) threading): two host threads, each with its own batch,
    tick the same audio at the same time; both must produce exactly what one handle produces alone"""
    import threading
    cfg = wf.Config.defaults(fft_size=2048, stereo=1, slope=1.0, bars=1, interp_mode=wf.INTERP["lanczos"])
    streams, ticks, hop = 256, 40, 800
    audio = [synth.block(SEED, 0, streams, 2, t * hop, hop) for t in range(ticks)]

    def run(out, idx):
        with wf.SpectrumBatch(cfg, streams) as b:
            for t in range(ticks):
                b.push_audio(audio[t])
                b.tick()
            out[idx] = (b.decibels(), b.bars(), b.last_silent())

    alone = [None]
    run(alone, 0)
    res = [None, None]
    th = [threading.Thread(target=run, args=(res, i)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for i in range(2):
        assert res[i] is not None, "a worker thread died"
        for got, want in zip(res[i], alone[0]):
            assert np.array_equal(got, want), f"handle {i} on its own thread differs from the single-threaded run"


@pytest.mark.parametrize("kind", ["spectrum_normalize", "meter"])
def test_sample_counters_wrap_at_2_to_the_32(kind):
    """a day of audio at 48 kHz overflows the 32-bit sample counters: a batch whose counters stand just below 2^32
    (wf_hip_debug_age) must go on producing exactly what a fresh batch produces from the same audio, across the wrap.
    In a child process on the development build of the library (tests/wrap_child.py): the release library has no test aids."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, WF_HIP_LIB=str(root / "waveform_amd" / "libwaveform_hip_dev.so"))
    r = subprocess.run([sys.executable, str(root / "tests" / "wrap_child.py"), kind], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "wrapped ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.parametrize("n,streams,extra", [(16384, 2048, dict(bars=1, interp_mode=1)), (8192, 2048, dict(bars=1, interp_mode=2)),
                                             (4096, 4096, dict(bars=1, interp_mode=1)), (4096, 2048, dict(curve=1, interp_mode=2, stereo=0)),
                                             (1024, 8192, dict(bars=1, interp_mode=1)), (512, 8192, dict(bars=1, interp_mode=1))])
def test_displays_are_deterministic_under_load(n, streams, extra):
    """the bars / curve tail shares the exchange buffer between wavefronts by an arrival counter and one barrier, and adds
    partial sums by lane shuffles: a race there would show as run-to-run differences under a chip full of workgroups.  Two
    runs of 150 back-to-back ticks (lanes on) must leave the same bits everywhere; the second run's ticks overlap differently
    (it starts on a busy device)."""
    kw = dict(fft_size=n, stereo=1, slope=1.0)
    kw.update(extra)
    cfg = wf.Config.defaults(**kw)
    ticks, hop = 150, 800
    res = []
    with wf.SpectrumBatch(cfg, streams, ring_frames=n + hop * 66) as warm:  # keeps the device busy before the second run
        warm.push_synth(SEED + 1, 0, hop * 64)
        for rep in range(2):
            with wf.SpectrumBatch(cfg, streams, ring_frames=n + hop * 66) as b:
                b.push_synth(SEED, 0, hop * 64)
                if rep:
                    warm.time_ticks(200, hop, hop * 63)
                b.time_ticks(ticks, hop, hop * 63)
                res.append((b.bars(), b.decibels(), b.last_silent()))
    for a, c, what in zip(res[0], res[1], ("bars", "rows", "silence flags")):
        assert np.array_equal(a, c), f"N={n}: {what} differ between two identical runs"
    assert np.all(np.isfinite(res[0][0]))


@pytest.mark.parametrize("n", [1024, 4096, 16384])
def test_starved_tick_in_the_middle_of_a_mono_mixdown(n):
    """WF_HIP_STARVED in the middle of a stream (the host's buffers hold less than window + A/V-sync delay: every channel is
    skipped, reference src/source_generic.cpp:55-61) with mono mixdown and volume normalisation: the end-of-tick pass still
    runs (:138-179) and mixes row 0's stale dB values with channel 1's last smoothed magnitudes, which the reference keeps
    linear in m_decibels[1] (:150-154).  With a gain that lifts row 0 above 0 dB the sum is positive and survives dbfs():
    the device must reproduce it (from channel 1's smoothing state), tick for tick, and recover afterwards."""
    cfg = wf.Config.defaults(fft_size=n, stereo=0, capture_channels=2, slope=1.0, normalize_volume=1, volume_target=-3.0, max_gain=45.0)
    hop, streams = 800, 3
    o = restate.OracleSource(cfg)
    o.set_input_rms(0.004)  # -48 dBFS measured -> the full 45 dB of gain: most of row 0 ends up above 0 dB
    with wf.SpectrumBatch(cfg, streams) as b:
        for t in range(10):
            a = synth.block(SEED, 0, 1, 2, t * hop, hop)
            starved = t == 5  # (one tick: a second one would feed on the first one's ill-conditioned output)
            if not starved:
                b.push_audio(np.broadcast_to(a, (streams, 2, hop)))
                o.push_audio(a[0])
            b.set_hidden(np.full(streams, 4 if starved else 0, np.uint8))  # WF_HIP_STARVED
            o.set_sync_delay(10 ** 7 if starved else 0)                     # the restatement underflows the reference's way
            b.tick(input_rms=0.004)
            o.tick(1.0 / 60.0)
            got = b.decibels()
            assert np.array_equal(got[0], got[1]) and np.array_equal(got[0], got[2])
            if t == 5:
                assert np.any(o.decibels()[0] > wf.db_min() + 1), "the scenario must leave values above DB_MIN on the starved tick"
            want = o.decibels()[:1]
            if starved:
                # dbfs((dB0 + mag1) / 2): a sum of a dB value of order 10 and a magnitude of order 1, both good to 1e-5 relative
                # -- where the two nearly cancel, the dB of the sum is as ill-conditioned as it is meaningless.  Held in the
                # domain of the sum itself: 1e-5 of the operands' size (|dB0| <= ~60) is 6e-4
                lg, lw = 10.0 ** (got[0][:1].astype(np.float64) / 20), 10.0 ** (want.astype(np.float64) / 20)
                assert np.all(np.abs(lg - lw) <= 1e-3 + 1e-5 * lw), f"N={n} tick {t} (starved): max {np.abs(lg - lw).max():.3e} in the mixed sum"
            else:
                assert_db_close(got[0][:1], want, f"N={n} tick {t}", deep=True)


@pytest.mark.parametrize("n,loud,quiet", [(4096, 2000.0, 1e-24), (65536, 2000.0, 1e-24), (16384, 2000.0, 1e-24), (800, 2000.0, 1e-24), (30000, 8.0, 1e-16)])
def test_magnitude_range_headroom_and_floor(n, loud, quiet):
    """|X|^2 = re^2 + im^2 is the one place where the path squares.  The device's window tables carry 2^40 up to 4096 samples, a
    power of two less per doubling beyond (2^24 on the Bluestein path through device memory) so that the square neither
    underflows for very quiet frames -- the reference's hypotf answers for the whole float range; here down to |X| ~ 1e-30 -- nor
    overflows below an amplitude of 4096 (+72 dBFS) at any size.  Audio scaled by a power of two far outside [-1, 1] in both directions must give
    the oracle's rows, shifted by exactly 20 log10 of the factor."""
    cfg = wf.Config.defaults(fft_size=n, stereo=1, slope=0.0, tsmoothing=wf.TSMOOTH["none"])
    hop, ticks = 800, 3
    blocks = [synth.block(SEED, 0, 1, 2, t * hop, hop)[0] for t in range(ticks)]
    for factor in (1.0, loud, quiet):
        g = np.float32(2.0 ** np.round(np.log2(factor)))   # a power of two: the scaling is exact
        o = restate.OracleSource(cfg)
        with wf.SpectrumBatch(cfg, 2) as b:
            for t in range(ticks):
                a = blocks[t] * g
                b.push_audio(np.broadcast_to(a, (2, 2, hop)))
                b.tick()
                o.feed_and_tick(a)
            got = b.decibels()[1]
        assert np.all(np.isfinite(got)), f"N={n} factor {g}: non-finite rows"
        assert_db_close(got, o.decibels(), f"N={n}, audio scaled by {g}", deep=True)
        if factor != 1.0:
            assert np.median(got) > wf.db_min() + 50, "the scaled frame must not have collapsed to DB_MIN"


def test_waveform_batch_with_per_stream_timestamps_and_paused_streams():
    """wf_hip_set_stream_audio_ts / wf_hip_set_stream_delay / WF_HIP_PAUSED on a waveform batch (ABI 9; what the plugin's batched
    waveform mode hands over): three streams that receive different amounts of audio per frame, each with its own end-of-audio
    timestamp and A/V-sync reserve, one of them sitting frames out -- every stream must equal a single-stream handle that was
    driven with the same values through wf_hip_tick_params, bit for bit."""
    import waveform_amd as wf
    from tools import synth
    cfg = wf.Config.defaults(waveform=1, stereo=1, width=640, meter_ms=100)
    sr = 48000
    hops = [(800, 441, 1024), (800, 800, 37), (0, 960, 800), (800, 441, 1024), (1024, 0, 800), (800, 800, 800), (441, 441, 441), (800, 0, 0)]
    delays = (0, 240, 960)
    paused_frames = {(2, 2), (2, 5)}  # (stream, frame): not ticked, nothing pushed
    with wf.SpectrumBatch(cfg, 3) as b, wf.SpectrumBatch(cfg, 1) as s0, wf.SpectrumBatch(cfg, 1) as s1, wf.SpectrumBatch(cfg, 1) as s2:
        singles = (s0, s1, s2)
        pos, ts = [0, 0, 0], [0, 0, 0]
        b.set_stream_delay(np.array(delays, np.uint32))
        for f, hop in enumerate(hops):
            state = np.zeros(3, np.uint8)
            for i in range(3):
                if (i, f) in paused_frames:
                    state[i] = 3  # WF_HIP_PAUSED
                    continue
                if hop[i]:
                    a = synth.block(SEED, 40 + i, 1, 2, pos[i], hop[i])
                    pos[i] += hop[i]
                    ts[i] = 5_000_000_000 + pos[i] * 1_000_000_000 // sr + 7 * i
                    b.push_audio(a, first=i)
                    singles[i].push_audio(a)
            b.set_hidden(state)
            b.set_stream_audio_ts(np.array(ts, np.uint64))
            b.tick(seconds=1 / 60)
            rows, silent = b.decibels(), b.last_silent()
            for i in range(3):
                if (i, f) not in paused_frames:
                    singles[i].tick(seconds=1 / 60, delay_frames=delays[i], audio_ts_ns=ts[i])
                assert np.array_equal(rows[i], singles[i].decibels()[0]), f"frame {f} stream {i}: rows differ from the single-stream handle"
                assert silent[i] == singles[i].last_silent()[0], f"frame {f} stream {i}: m_last_silent"
                assert b.waveform_ts(i, 1)[0] == singles[i].waveform_ts()[0], f"frame {f} stream {i}: m_waveform_ts"
    mcfg = wf.Config.defaults(meter=1)
    with wf.SpectrumBatch(mcfg, 2) as m:
        with pytest.raises(wf.WfHipError):
            m.set_stream_audio_ts(np.zeros(2, np.uint64))   # not a waveform batch
        with pytest.raises(wf.WfHipError):
            m.waveform_ts()


def test_committed_profiles_are_of_the_kernels_that_run():
    """bench.py replays roofline.traffic and the trace span from the newest committed profiles/rNN*_<shape>_pmc.json, but only when
    its `kernel` string equals wf_hip_kernel_name() of the handle that runs (a summary of a kernel that no longer runs is not
    replayed: traffic null).  For every shape bench.py names: a handle created with that shape's configuration must report the
    kernel the newest committed summary was taken on -- a kernel change without a re-profile fails HERE instead of dropping
    roofline.traffic from the driver's line."""
    import json
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root))
    import bench
    ema = dict(stereo=1, slope=1.0, window=wf.WINDOW["hann"], tsmoothing=wf.TSMOOTH["exponential"], gravity=0.65)
    shapes = [("cfg3_n4096", wf.Config.defaults(fft_size=4096, **ema), 4096)] + [(s[5], s[1], s[2]) for s in bench.shape_list(wf)]
    for key, cfg, streams in shapes:
        with wf.SpectrumBatch(cfg, streams, ring_frames=cfg.fft_size + 1600) as b:
            live = b.kernel_name()
        files = sorted((root / "profiles").glob(f"r*_{key}_pmc.json"))
        assert files, f"{key}: no committed summary"
        d = json.loads(files[-1].read_text())
        assert d.get("kernel") == live, f"{key}: {files[-1].name} was taken on '{d.get('kernel')}', the library runs '{live}': re-profile (tools/gpu_round.sh)"
        assert bench.pmc_profile(key, live) is not None
