#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X spectrum path (contract: see the task brief).

Metric (BASELINE.json): spectra/sec at FFT=4096, batch=4096 streams, with the achieved HBM
GB/s of the fused kernel against the chip's peak.

Workload at every N: BASELINE.json configs[2] per GPU -- 4096 independent stereo streams
(8192 spectra per tick), FFT 4096, Hann window, EXPONENTIAL smoothing g=0.65, slope 1.0,
48 kHz synthetic white noise (include/wf_synth.h), hop 800 samples (60 fps).  Weak scaling:
every rank owns its own 4096 streams, no data-path collective (the streams share nothing,
SURVEY.md section 8(e)).

A "step" is one tick_spectrum over the whole batch.  All audio for warm-up + timed ticks is
generated into the device rings before the timed region (inputs resident in HBM); tick i
analyses the window that ends (steps-1-i)*hop frames before the newest sample, i.e. exactly
the ring contents the reference would see at that video frame (its own A/V-sync path,
src/source_generic.cpp:50-59).

`python bench.py --gpus N` launches itself: with WORLD_SIZE unset and N > 1 it re-executes under
torch.distributed.run (one rank per GPU, 127.0.0.1 rendezvous), capped at the devices the box
has -- the line says how many were measured.

One JSON line on rank 0.  Extra objects:
  roofline       algorithmic bytes per tick / the wall-clock time per step of the timed region vs 8 TB/s HBM peak
                 (frac); frac_events (HIP events on the library's stream over the same region), frac_trace (the
                 committed rocprofv3 kernel trace named in `profile`), cold (the same steps with no lead-in)
  cpu_baseline   the reference's own classes (oracle/_ref/libwfref.so: verbatim TUs + vendored FFTW) timed on
                 this box's host cores on a bounded sample (rank 0, N=1 only): WAVSourceAVX2 (value; 1 thread
                 and all cores) and WAVSourceGeneric (the parity target); every other_configs shape carries
                 the same at its own fft size
  other_configs  (N=1) the same measurement on the other shapes the north star names: working sets
                 past the 256 MB Infinity Cache (8192 / 16384 streams), BASELINE configs[3]
                 (N=16384 x 1024 streams, TV-EMA + Lanczos bars), configs[1], the configs[4] per-GPU
                 shape (8192 streams, bars only), each with its own roofline object
  pcie_inclusive (N=1) the headline shape fed through the host boundary every step (page-locked
                 buffers, wf_hip_push_audio_async under the previous tick) -- never `value`
  configs4       (every N) BASELINE configs[4] in the same process group: 8192 stereo streams per rank, bars-only
                 ticks, ONE all-gather of the bar heights per tick (RCCL when N > 1) under the next tick, verified
                 by per-rank checksums; its own ms_per_step / roofline / per-rank device times
  c_abi_multi    (N=1) the same configs[4] shape through the C ABI's single-process multi-device group
                 (wf_hip_multi_*: one host thread per device, dlopen()ed RCCL) over every device this process sees
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
FFT_SIZE = 4096
STREAMS_PER_GPU = 4096
HOP = 800
SEED = 0x5741564546524D31
MAX_DEPTH = 64   # ticks of audio resident per stream (--depth); the walk starts over when it reaches the newest sample (the same
                 # work per step, no host synchronisation inside the timed region).  The depth does not matter: 16 / 64 / 512
                 # measure 0.79-0.80 of peak alike (r02j)


CFG4_WATCHDOG_S = 240.0  # multi-rank runs: the configs[4] region (a few seconds when healthy) may take this long at most


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=500, help="untimed steps first: the device needs ~300 ticks (15-20 ms) of load before its "
                    "clocks have settled -- 20 warm-up ticks measure 0.72-0.73 of peak, 300 and more 0.78-0.79 (profiles/r02j_warmup.txt)")
    ap.add_argument("--streams", type=int, default=STREAMS_PER_GPU, help="streams per GPU")
    ap.add_argument("--fft", type=int, default=FFT_SIZE)
    ap.add_argument("--lead-in-ms", type=float, default=40.0,
                    help="device time of untimed ticks in front of the warm-up steps (clock settling; 0: none)")
    ap.add_argument("--depth", type=int, default=MAX_DEPTH,
                    help="ticks of audio resident per stream before the timed region (ring = window + depth hops, rounded up to a power of two)")
    ap.add_argument("--bars-allgather", action="store_true",
                    help="BASELINE configs[4] shape: bars (26 Lanczos bars per channel) computed in the tick (bars-only mode) and "
                         "all-gathered across ranks (RCCL over xGMI) under the next tick; changes the workload, so it is off by default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the other_configs / pcie_inclusive legs (profiling runs)")
    ap.add_argument("--cpu-seconds", type=float, default=30.0, help="CPU work budget (core-seconds) of the baseline leg")
    return ap.parse_args()


def cpu_baseline(fft: int, cores: int, budget_core_s: float, settings: dict | None = None, classes=("avx2", "generic")):
    """The reference's own CPU classes on the host cores (oracle/_ref/libwfref.so: verbatim TUs + vendored FFTW), bounded sample:
    WAVSourceAVX2 (the headline baseline: `value`) and WAVSourceGeneric (the parity target), each on 1 thread and on `cores`
    threads.  budget_core_s: core-seconds of CPU work for the T-thread run of each class."""
    from oracle import wfref
    if not wfref.available():
        return None
    settings = dict(settings or dict(enable_large_fft=True, channel_mode="stereo", slope=1.0, window="hann",
                                     temporal_smoothing="exp_moving_avg", gravity=0.65), fft_size=fft)
    out, streams_per_thread = {}, 8
    for isa in classes:
        # calibrate on one core, then size the sample to the budget
        v1, _ = wfref.bench(isa, settings, 4, 1, 8, 64, hop=HOP, seed=SEED)
        if v1 <= 0:
            continue
        per_core = max(budget_core_s / max(cores, 1), 0.25)          # seconds of work per thread
        ticks = int(max(32, min(16384, per_core * v1 / (2 * streams_per_thread))))
        v1, el1 = wfref.bench(isa, settings, streams_per_thread, 1, 8, max(32, ticks // 4), hop=HOP, seed=SEED)
        n_streams = streams_per_thread * cores
        v, el = wfref.bench(isa, settings, n_streams, cores, 16, ticks, hop=HOP, seed=SEED)
        out[isa] = {"value": v, "cores": cores, "value_1_thread": v1, "streams": n_streams, "ticks": ticks, "wall_s": el + el1}
    if "avx2" not in out:
        return None
    a = out["avx2"]
    res = {
        "value": a["value"], "unit": "spectra/s", "cores": cores, "kind": "reference", "class": "WAVSourceAVX2",
        "value_1_thread": a["value_1_thread"],
        "sample": f"WAVSourceAVX2::tick_spectrum (verbatim reference + vendored FFTW 3.3.11), {a['streams']} stereo streams x "
                  f"{a['ticks']} ticks, hop {HOP}, FFT {fft}, {cores} threads, {a['wall_s']:.2f} s wall; 1 thread: {a['value_1_thread']:.0f} spectra/s",
    }
    if "generic" in out:
        g = out["generic"]
        res["generic"] = {"value": g["value"], "value_1_thread": g["value_1_thread"], "cores": cores, "class": "WAVSourceGeneric (the parity target)",
                          "sample": f"{g['streams']} stereo streams x {g['ticks']} ticks, {g['wall_s']:.2f} s wall"}
    return res


def shape_settings(wf, cfg):
    """the reference plugin's setting keys for a wf_config (what tests/helpers.ref_settings does; kept here so that bench.py imports
    nothing from tests/)"""
    win = {v: k for k, v in wf.WINDOW.items()}[int(cfg.window)]
    ts = {0: "none", 1: "exp_moving_avg", 2: "tv_exp_moving_avg"}[int(cfg.tsmoothing)]
    return dict(enable_large_fft=True, auto_fft_size=False, channel_mode="stereo" if cfg.stereo else "mono", window=win,
                temporal_smoothing=ts, gravity=repr(float(cfg.gravity)), slope=repr(float(cfg.slope)), fast_peaks=bool(cfg.fast_peaks))


def host_cores() -> int:
    """CPU threads this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) or 1
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def pmc_profile(shape: str, kernel: str):
    """The latest committed rocprofv3 summary of this shape (profiles/rNN*_<shape>_pmc.json) that was taken on the kernel this
    run executes (its `kernel` string must equal wf_hip_kernel_name()), or None -- a summary of a kernel that no longer runs
    is not replayed."""
    best = None
    for p in sorted((ROOT / "profiles").glob(f"r*_{shape}_pmc.json")):
        try:
            d = json.loads(p.read_text())
        except Exception:
            continue
        if d.get("hbm_bytes_per_launch") and d.get("kernel") == kernel:
            best = (p.name, d)
    return best


def roofline(batch, shape, kernel_ms, flags=0, wall_ms=None, cold_ms=None):
    """frac: algorithmic bytes per tick / the barrier-bracketed WALL-CLOCK time per step of this run (host launch overhead and the
    final synchronisation included) -- the figure the driver's own clock reproduces.
    frac_events: the same bytes / device time per tick (HIP events on the library's stream around the same region): a few
    percent above frac.
    frac_trace: the same bytes / tick_span_ns of the committed rocprofv3 kernel trace named in `profile` (profiler attached).
    traffic is NOT measured in this run (PMC counters need rocprofv3 passes of their own): it is replayed from the committed
    summary named in traffic_source, which must have been taken on the kernel this run executes."""
    algo = batch.algorithmic_bytes_per_tick(flags)
    ev = algo / (kernel_ms * 1e-3) / 1e9
    achieved = algo / (wall_ms * 1e-3) / 1e9 if wall_ms else ev
    prof = pmc_profile(shape, batch.kernel_name()) if shape else None
    out = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
           "timed_by": "wall clock per step (barrier + synchronize on both sides)" if wall_ms else "HIP events",
           "frac_events": ev / HBM_PEAK_GBPS, "achieved_events": ev,
           "frac_wall": (achieved / HBM_PEAK_GBPS) if wall_ms else None,
           "traffic": (prof[1].get("hbm_bytes_per_tick_all_kernels") or prof[1].get("hbm_bytes_per_tick") or prof[1]["hbm_bytes_per_launch"]) if prof else None,
           "traffic_source": (f"profiles/{prof[0]} (committed rocprofv3 --pmc passes of this shape and kernel; not measured in this run)" if prof else None),
           "kernel": batch.kernel_name(), "kernel_ms": kernel_ms, "algorithmic_bytes_per_tick": algo,
           "algorithmic_bytes_per_kernel_launch": algo // max(batch.launches_per_tick(), 1),
           # A tick goes out as kernel_launches_per_tick launches (lanes on their own HIP streams run concurrently), timed
           # together by the events (kernel_ms).  rocprofv3's per-launch average is one slice
           # (algorithmic_bytes_per_kernel_launch) sharing the chip with the others; what must agree with kernel_ms is the tick
           # span of its kernel trace (profile.tick_span_ms)
           "kernel_launches_per_tick": batch.launches_per_tick()}
    if cold_ms:
        out["cold"] = {"ms_per_step": cold_ms, "frac": algo / (cold_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                       "what": "the same steps timed the same way as the first thing the process does on the device (no lead-in: the clocks are still ramping)"}
    if prof:
        tr, ks = prof[1].get("trace") or {}, prof[1].get("kernel_stats") or {}
        span = tr.get("tick_span_ns") or 0
        out["profile"] = {"file": "profiles/" + prof[0], "tick_span_ms": span * 1e-6 or None,
                          "avg_launch_ms": (ks.get("avg_ns") or 0) * 1e-6 or None, "launches_per_tick": tr.get("launches_per_tick")}
        out["frac_trace"] = (algo / (span * 1e-9) / 1e9 / HBM_PEAK_GBPS) if span else None
    return out


def copy_ceiling(torch, mib=1024, reps=20):
    """What a plain device-to-device copy reaches on this GPU, in this process, right now (torch's elementwise copy kernel on `mib`
    MiB, read + written bytes / time by events): the memory system's own efficiency.  /opt/skills/guides/MI355X_MICROARCH.md
    quotes 6.29 TB/s (79 % of the 8 TB/s peak) for a float4 copy; `frac` everywhere in this line stays against the 8 TB/s peak."""
    n = mib * (1 << 20) // 4
    src = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
    dst = torch.empty_like(src)
    for _ in range(5):
        dst.copy_(src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    gbps = 2.0 * n * 4 * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del src, dst
    return {"achieved": gbps, "unit": "GB/s", "frac_of_peak": gbps / HBM_PEAK_GBPS, "what": f"torch copy_ of {mib} MiB, bytes read + written, {reps} repetitions by events"}


WARM_MS, TIMED_MS = 40.0, 30.0  # other shapes: device time of the untimed lead-in (clocks settle in 15-20 ms) and of the timed region


def measure_shape(wf, name, cfg, streams, steps, warmup, device, flags=0, shape=None):
    """One of the other shapes: back-to-back ticks over resident audio (64 ticks of it, walked again and again); a lead-in
    of at least `warmup` ticks and WARM_MS of device time, then at least `steps` ticks and TIMED_MS; wall clock and device events."""
    depth = 64
    with wf.SpectrumBatch(cfg, streams, device=device, ring_frames=cfg.fft_size + HOP * (depth + 1)) as b:
        b.push_synth(SEED, 0, HOP * depth)
        b.sync()
        first = HOP * (depth - 1)
        probe = b.time_ticks(16, HOP, first, flags=flags)  # ms per tick, cold
        warm = max(warmup, int(WARM_MS / probe) + 1)
        steps = max(steps, int(TIMED_MS / probe) + 1)
        b.time_ticks(warm, HOP, first, flags=flags)
        t0 = time.perf_counter()
        ms = b.time_ticks(steps, HOP, first, flags=flags)
        wall = time.perf_counter() - t0
        spectra = streams * b.capture_channels
        return {"name": name, "streams": streams, "fft_size": int(cfg.fft_size), "spectra_per_tick": spectra, "steps": steps, "warmup": warm + 16,
                "value": spectra * steps / wall, "unit": "spectra/s", "ms_per_step": wall * 1e3 / steps,
                "roofline": roofline(b, shape, ms, flags, wall_ms=wall * 1e3 / steps)}


def shape_list(wf):
    ema = dict(stereo=1, slope=1.0, window=wf.WINDOW["hann"], tsmoothing=wf.TSMOOTH["exponential"], gravity=0.65)
    # BASELINE shapes first (the driver keeps the head of the line's standard keys and the last 8 KB of stdout), the reference-range extras last
    shapes = [
        ("configs[3]: 1024 stereo streams, FFT 16384, TV-EMA (gravity) + 26 Lanczos bars per channel",
         wf.Config.defaults(fft_size=16384, stereo=1, window=wf.WINDOW["hann"], tsmoothing=wf.TSMOOTH["tvexponential"], gravity=0.65,
                            bars=1, interp_mode=wf.INTERP["lanczos"]), 1024, 60, 0, "cfg4_n16384_bars"),
        ("configs[4] per-GPU shape: 8192 stereo streams, FFT 4096, EMA + slope, 26 Lanczos bars per channel, bars only (no m_decibels store)",
         wf.Config.defaults(fft_size=4096, bars=1, interp_mode=wf.INTERP["lanczos"], **ema), 8192, 60, wf.TICK_NO_DECIBELS, "cfg5shape_8192streams_barsonly"),
        ("configs[1] as a batch: 256 stereo streams, FFT 2048, Hann + magnitude + dB, no smoothing",
         wf.Config.defaults(fft_size=2048, stereo=1, window=wf.WINDOW["hann"], tsmoothing=wf.TSMOOTH["none"]), 256, 60, 0, "cfg2_batch"),
        ("configs[2] x2: 8192 stereo streams, FFT 4096, EMA + slope (536 MB working set, past the 256 MB Infinity Cache)",
         wf.Config.defaults(fft_size=4096, **ema), 8192, 60, 0, "cfg3_8192streams"),
        ("configs[2] x4: 16384 stereo streams, FFT 4096, EMA + slope (1.07 GB working set)",
         wf.Config.defaults(fft_size=4096, **ema), 16384, 40, 0, "cfg3_16384streams"),
        # the ends of the reference's FFT range (not BASELINE configs; reported so that the driver's line carries them too)
        ("fft_size 65536 (the reference's maximum, 'enable large FFT'): 256 stereo streams, EMA + slope; both rows of 16384 complex points and "
         "the end of the tick in one workgroup", wf.Config.defaults(fft_size=65536, **ema), 256, 30, 0, "n65536"),
        ("fft_size 800 (the plugin's automatic size at 48 kHz / 60 fps; 400 complex points by a mixed-radix plan -- the radices are in the kernel name): 8192 stereo streams, EMA + slope",
         wf.Config.defaults(fft_size=800, **ema), 8192, 60, 0, "n800_mixed_radix"),
    ]
    return shapes


def other_configs(wf, device, cpu_cores=0, cpu_core_s=4.0):
    """cpu_cores > 0: every shape also carries the reference's CPU classes at ITS fft size and settings, a
    few core-seconds each, once per distinct (fft size, smoothing) pair"""
    out, cpu_seen = [], {}
    for name, cfg, streams, steps, flags, shape in shape_list(wf):
        try:
            r = measure_shape(wf, name, cfg, streams, steps, 8, device, flags, shape)
            r["shape"] = shape
            if cpu_cores > 0:
                key = (int(cfg.fft_size), int(cfg.tsmoothing))
                if key not in cpu_seen:
                    try:
                        cpu_seen[key] = cpu_baseline(int(cfg.fft_size), cpu_cores, cpu_core_s, shape_settings(wf, cfg))
                    except Exception as e:
                        cpu_seen[key] = {"error": str(e)}
                r["cpu_baseline"] = cpu_seen[key]
            out.append(r)
        except Exception as e:  # reported, never required
            out.append({"name": name, "error": str(e)})
    return out


def pcie_inclusive(wf, cfg, streams, device, steps=200, warm=400):
    """The headline shape with every step's audio crossing the host boundary: one 60 fps hop per stream in page-locked
    memory -> wf_hip_push_audio_async (H2D on the copy stream under the previous tick) -> ring append -> tick.
    Two things slow these copies from 54 to 37-45 GB/s and are kept out of the figure: another batch alive in the process
    (45 GB/s for as long as it lives), and the time after a batch has been closed -- 0.15 s after the headline's 2 GB of
    rings, 0.65 s after the 13 GB of the larger shapes, as if the freed memory were being cleared by the same DMA engines.
    Hence: after the headline batch is closed, before the other shapes, behind 400 untimed steps (0.2 s)."""
    import numpy as np
    from tools import synth
    packet = np.ascontiguousarray(np.broadcast_to(synth.block(SEED, 0, 1, 2, 0, HOP), (streams, 2, HOP)), np.float32)
    with wf.SpectrumBatch(cfg, streams, device=device) as b:
        pin = [wf.PinnedBuffer(packet.shape), wf.PinnedBuffer(packet.shape)]
        for p in pin:
            p.array[...] = packet
        for i in range(warm + steps):
            if i == warm:
                b.sync()
                t0 = time.perf_counter()
            slot = i & 1
            b.ingest_done(slot)  # the buffer is free again (a live host refills it here)
            b.push_audio_async(pin[slot], streams, HOP, slot)
            b.tick()
        b.sync()
        dt = (time.perf_counter() - t0) / steps
        for p in pin:
            p.close()
    return {"value": streams * 2 / dt, "unit": "spectra/s", "ms_per_step": dt * 1e3, "host_GBps": packet.nbytes / dt / 1e9,
            "path": "page-locked host buffer -> wf_hip_push_audio_async (H2D under the previous tick) -> ring append -> tick",
            "bytes_per_step": int(packet.nbytes), "steps": steps}


CFG4_STREAMS_PER_GPU = 8192  # BASELINE configs[4]: 65536 concurrent streams over 8 GPUs


def configs4_region(wf, torch, dist, rank, world, local_rank, steps, lead_in_ms=40.0):
    """BASELINE configs[4] per GPU, in the same process group as the headline: 8192 stereo streams per rank (65536 over 8
    GPUs), FFT 4096, EMA + slope, 26 Lanczos bars per channel, bars-only ticks, and after every tick ONE all-gather of
    [streams/rank][2][26] bar heights (RCCL over xGMI when world > 1) issued on a side stream under the next tick
    (waveform_amd.dist.BarsGather).  Timed like the headline: barrier + synchronize on both sides, max over ranks.
    Verified: every rank's block of the final gathered result must carry the checksum that rank computed over its own bars
    (exact integer sums of the float bit patterns, all-gathered alongside)."""
    from waveform_amd.dist import shard_streams, BarsGather
    cfg = wf.Config.defaults(fft_size=FFT_SIZE, stereo=1, slope=1.0, window=wf.WINDOW["hann"], tsmoothing=wf.TSMOOTH["exponential"],
                             gravity=0.65, bars=1, interp_mode=wf.INTERP["lanczos"])
    flags = wf.TICK_NO_DECIBELS
    streams, depth = CFG4_STREAMS_PER_GPU, 16
    total = streams * world
    shard = shard_streams(total, rank, world)
    batch = wf.SpectrumBatch(cfg, streams, device=local_rank, ring_frames=FFT_SIZE + HOP * (depth + 1))
    try:
        batch.push_synth(SEED, 0, HOP * depth, stream_id0=shard.first)
        batch.sync()
        gather = BarsGather(batch, shard)

        def barrier():
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()

        def run(n):
            batch.time_begin()
            for i in range(n):
                batch.tick(delay_frames=HOP * (depth - 1 - i % depth), flags=flags)
                gather.launch()
            ms = batch.time_end()
            gather.wait()
            return ms / n

        probe = run(8)
        warm = 8 + int(lead_in_ms / max(probe, 1e-4)) + 1
        if dist is not None:
            # every tick carries a collective: the ranks must agree on the number of ticks, and each derived its own from its
            # own clock (a count that differs by one between two ranks leaves one of them in an all-gather nobody answers)
            w = torch.tensor([warm], dtype=torch.int64, device="cuda")
            dist.all_reduce(w, op=dist.ReduceOp.MAX)
            warm = int(w.item())
        run(warm - 8)
        barrier()
        t0 = time.perf_counter()
        kernel_ms = run(steps)
        barrier()
        elapsed = time.perf_counter() - t0
        # verification: the gathered result of the last tick, block by block, against what each rank holds itself
        full = gather.wait()                                             # [total][2][26] on this rank
        own = torch.empty((streams, batch.display_channels, batch.num_bars), dtype=torch.float32, device="cuda")
        side = torch.cuda.Stream()  # (the default stream's handle is 0: the library takes a real hipStream_t)
        batch.copy_bars_to_device_async(own.data_ptr(), side.cuda_stream)
        side.synchronize()
        from waveform_amd.dist import verify_gathered
        verified = verify_gathered(full, own, shard)   # (collective when world > 1: every rank checks the copy it received)
        times = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device="cuda")
        if dist is not None:
            tl = [torch.empty_like(times) for _ in range(world)]
            dist.all_gather(tl, times)
            tl = torch.stack(tl).cpu().numpy()
        else:
            tl = times.cpu().numpy()[None]
        finite = bool(torch.isfinite(full).all().item())
        wall = float(tl[:, 0].max())
        dev_ms = [float(x) for x in tl[:, 1]]
        spectra = streams * batch.capture_channels
        out = {
            "name": f"BASELINE configs[4]: {total} concurrent stereo streams sharded over {world} GPU(s) ({streams} per GPU), FFT {FFT_SIZE}, "
                    "EMA + slope, 26 Lanczos bars per channel, bars-only ticks; all-gather of the bar heights under the next tick",
            "streams_total": total, "streams_per_gpu": streams, "spectra_per_tick": spectra * world, "steps": steps, "warmup": warm,
            "value": spectra * world * steps / wall, "unit": "spectra/s", "ms_per_step": wall * 1e3 / steps,
            "device_ms_per_tick": {"min": min(dev_ms), "max": max(dev_ms), "per_rank": dev_ms},
            "collective": (f"all_gather_into_tensor (backend {dist.get_backend()}{': RCCL' if dist.get_backend() == 'nccl' else ''})" if world > 1
                           else "none (world 1: the local device copy of the same path)"),
            "gathered_bytes_per_rank_per_tick": int(streams * batch.display_channels * batch.num_bars * 4),
            "gathered_bytes_total_per_tick": int(total * batch.display_channels * batch.num_bars * 4),
            "verified": verified and finite,
            "verification": "per-rank position-weighted integer checksum of the float bit patterns of its own bars == checksum of its block in every rank's gathered copy; all values finite",
            "roofline": roofline(batch, "cfg5shape_8192streams_barsonly", max(dev_ms), flags, wall_ms=wall * 1e3 / steps),
        }
    finally:
        batch.close()
    return out


def c_abi_multi(timeout_s=240):
    """BASELINE configs[4] through the C ABI's own multi-device group (wf_hip_multi_*: single process, one host thread per
    device, ncclAllGather of a dlopen()ed librccl.so) over every device this process can see -- tools/multi_bench.py in a child
    process (no torch there; a wedged RCCL costs the child, not the line)."""
    try:
        r = subprocess.run([sys.executable, str(ROOT / "tools" / "multi_bench.py")], capture_output=True, text=True, timeout=timeout_s)
        lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": f"rc {r.returncode}: {r.stderr[-400:]}"}
        return json.loads(lines[-1])
    except subprocess.TimeoutExpired:
        return {"error": f"timed out after {timeout_s} s"}
    except Exception as e:
        return {"error": str(e)}


def self_launch(args):
    """--gpus N without a launcher: re-execute under torch.distributed.run, one rank per GPU this box has."""
    import torch
    have = torch.cuda.device_count()
    n = min(args.gpus, have)
    if n < 1:
        print("bench.py: no GPU visible (torch.cuda.device_count() is 0)", file=sys.stderr)
        sys.exit(3)
    if n < args.gpus:
        print(f"bench.py: --gpus {args.gpus} requested, {have} device(s) present: measuring {n}", file=sys.stderr)
    if n == 1:
        return 1  # run in this process
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    argv = [a for a in sys.argv[1:]]
    # replace the --gpus value with what is measured
    out = []
    skip = False
    for a in argv:
        if skip:
            skip = False
            continue
        if a == "--gpus":
            skip = True
            continue
        if a.startswith("--gpus="):
            continue
        out.append(a)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), "--gpus", str(n)] + out
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.run(cmd, env=env).returncode)


def main():
    args = parse_args()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        args.gpus = self_launch(args)  # returns only when a single device is measured in this process
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test aid (tests/test_gpu_fullsize.py): the multi-rank code path on a box with fewer devices than ranks -- the ranks share
    # the devices there are and the collectives run on gloo (RCCL refuses two ranks on one device).  Never set by the driver.
    share = os.environ.get("WF_BENCH_SHARE_DEVICES") == "1"
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch
    import waveform_amd as wf

    if not torch.cuda.is_available():
        print("bench.py: no GPU visible (torch.cuda.is_available() is False)", file=sys.stderr)
        sys.exit(3)
    if share:
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    cfg = wf.Config.defaults(fft_size=args.fft, stereo=1, slope=1.0, window=wf.WINDOW["hann"],
                             tsmoothing=wf.TSMOOTH["exponential"], gravity=0.65)
    flags = 0
    if args.bars_allgather:
        cfg.bars = 1
        cfg.interp_mode = wf.INTERP["lanczos"]
        flags = wf.TICK_NO_DECIBELS
    total_ticks = args.warmup + args.steps
    depth = max(1, min(total_ticks, args.depth))
    ring_frames = args.fft + HOP * (depth + 1)
    batch = wf.SpectrumBatch(cfg, args.streams, device=local_rank, ring_frames=ring_frames)
    spectra_per_step = args.streams * batch.capture_channels

    # all audio resident before the timed region; every rank generates its own streams
    batch.push_synth(SEED, 0, HOP * depth, stream_id0=rank * args.streams)
    batch.sync()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    gather = None
    if args.bars_allgather:
        from waveform_amd.dist import shard_streams, BarsGather
        gather = BarsGather(batch, shard_streams(args.streams * world, rank, world))

    def run(n_ticks):
        """n_ticks steps over the resident audio, oldest window first (a walk that reaches the newest sample starts over, no
        host synchronisation in between); returns the average device time per tick in ms"""
        if gather is None:
            return batch.time_ticks(n_ticks, HOP, HOP * (depth - 1))
        # tick i, its bars handed to the gather stream (wf_hip_copy_bars_device_async), the all-gather of tick i under
        # tick i+1; nothing here waits on the host
        batch.time_begin()
        for i in range(n_ticks):
            batch.tick(delay_frames=HOP * (depth - 1 - i % depth), flags=flags)
            gather.launch()
        ms = batch.time_end()
        gather.wait()
        return ms / n_ticks

    # Lead-in: the device's clocks settle after 15-20 ms of load (profiles/r02j_warmup.txt), whatever --warmup says -- a caller
    # that passes a handful of warm-up steps would otherwise time the ramp.  Untimed ticks of the same batch until
    # --lead-in-ms of device time have passed (0 disables); then the W warm-up steps the contract asks for.
    # the cold figure: the same K steps, timed the same way, as the first thing this process does on the device
    cold_ms = None
    if args.lead_in_ms > 0 and gather is None:
        barrier()
        tc = time.perf_counter()
        run(args.steps)
        barrier()
        cold_ms = (time.perf_counter() - tc) * 1e3 / args.steps
    lead_in_ticks = 0
    if args.lead_in_ms > 0:
        probe = run(16)
        lead_in_ticks = 16 + int(args.lead_in_ms / max(probe, 1e-4)) + 1
        if dist is not None:  # (the same count on every rank: with --bars-allgather every tick carries a collective)
            w = torch.tensor([lead_in_ticks], dtype=torch.int64, device="cuda")
            dist.all_reduce(w, op=dist.ReduceOp.MAX)
            lead_in_ticks = int(w.item())
        run(lead_in_ticks - 16)
    # warm-up: W untimed steps
    if args.warmup > 0:
        run(args.warmup)
    barrier()
    t0 = time.perf_counter()
    # K timed steps, HIP events around them on the library's stream
    kernel_ms = run(args.steps)
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    per_rank_ms, ranks_seen = [elapsed * 1e3 / args.steps], 1
    # The scaling curve's N = 1 point, taken in THIS run: rank 0 alone over the same K steps while the other ranks wait at a
    # barrier (their GPUs idle) -- the driver computes efficiency itself from its own N = 1 run; this is the same ratio from one
    # process group, one box, one minute, and it travels in the line (config.scaling_point, and again at the line's very end).
    solo_rate = None
    if dist is not None and gather is None:
        if rank == 0:
            ts = time.perf_counter()
            run(args.steps)
            torch.cuda.synchronize()
            solo_rate = spectra_per_step * args.steps / (time.perf_counter() - ts)
        barrier()
    if dist is not None:
        # every rank's own time per step and device time per tick, and how many ranks the collective really reached
        mine = torch.tensor([elapsed * 1e3 / args.steps, kernel_ms, 1.0], dtype=torch.float64, device="cuda")
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        allr = torch.stack(allr).cpu()
        per_rank_ms = [float(x) for x in allr[:, 0]]
        per_rank_kernel_ms = [float(x) for x in allr[:, 1]]
        one = torch.ones(1, dtype=torch.float64, device="cuda")
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        ranks_seen = int(round(float(one.item())))
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        k = torch.tensor([kernel_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(k, op=dist.ReduceOp.MAX)
        kernel_ms = float(k.item())

    if dist is not None and ranks_seen != world:
        print(f"bench.py: rank {rank}: the process group has {world} ranks but an all-reduce over it reached {ranks_seen}", file=sys.stderr, flush=True)
        os._exit(4)
    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = spectra_per_step * world * args.steps / elapsed
        out = {
            "metric": "spectra/sec at FFT=4096, batch=4096 streams; achieved HBM GB/s vs peak",
            "value": value,
            "unit": "spectra/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "warmup_total": args.warmup + lead_in_ticks + (args.steps if cold_ms else 0),  # every untimed tick in front of the timed region: the cold region, the lead-in, the W requested
            "lead_in": {"ms": args.lead_in_ms, "ticks": lead_in_ticks, "why": "untimed ticks until the device's clocks have settled (15-20 ms of load), in front of the warm-up steps"},
            "ms_per_step": ms_per_step,
            # one process per GPU: what every rank measured itself (ms_per_step above is the slowest), the backend of the process
            # group and how many ranks an all-reduce over it reached -- a rank that silently dropped out shows here
            "ranks": {"world": world, "seen_by_all_reduce": ranks_seen, "backend": (dist.get_backend() if dist is not None else "none (one process)"),
                      "ms_per_step_per_rank": per_rank_ms, "device_ms_per_tick_per_rank": (per_rank_kernel_ms if dist is not None else [kernel_ms])},
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": (f"BASELINE configs[{4 if args.bars_allgather else 2}]{' shape (per GPU)' if args.bars_allgather else ''}: "
                             f"{args.streams} independent stereo streams per GPU ({spectra_per_step} spectra/tick), "
                             f"FFT={args.fft}, Hann, EMA g=0.65 + slope 1.0"
                             f"{', 26 Lanczos bars per channel (bars-only ticks) all-gathered under the next tick' if args.bars_allgather else ''}, "
                             f"48 kHz counter-hash white noise, hop {HOP}"),
                "streams_per_gpu": args.streams, "fft_size": args.fft, "hop": HOP,
                "parallelism": (f"streams sharded over {world} GPU(s); bar heights all-gathered after every step" if args.bars_allgather
                                else f"streams sharded over {world} GPU(s), no data-path collective"),
            },
            "roofline": roofline(batch, "cfg3_n4096" if (args.streams, args.fft, flags) == (STREAMS_PER_GPU, FFT_SIZE, 0) else None, kernel_ms, flags,
                                 wall_ms=ms_per_step, cold_ms=cold_ms),
        }
        n1 = solo_rate if solo_rate else value
        out["config"]["scaling_point"] = {"n": world, "spectra_per_s": value, "per_gpu_spectra_per_s": value / world, "n1_spectra_per_s": n1,
                                          "efficiency_vs_n1": value / (world * n1),
                                          "n1_how": ("rank 0 alone over the same steps, the other ranks idle at a barrier, same process group" if solo_rate
                                                     else "this run IS the N = 1 point")}
        # every rank's own fraction of its GPU's HBM peak -- by its wall clock per step and by its device time per tick -- next to the
        # job's (the slowest rank's): a GPU that lags the others on a node shows here
        algo_per_tick = out["roofline"]["algorithmic_bytes_per_tick"]
        peak = out["roofline"]["peak"]
        dev_ms_per_rank = per_rank_kernel_ms if dist is not None else [kernel_ms]
        out["roofline"]["frac_per_rank"] = [algo_per_tick / (ms * 1e-3) / 1e9 / peak for ms in per_rank_ms]
        out["roofline"]["frac_events_per_rank"] = [algo_per_tick / (ms * 1e-3) / 1e9 / peak for ms in dev_ms_per_rank]
        try:  # the memory system's own ceiling, measured here and now: the headline kernel against a plain copy
            cc = copy_ceiling(torch)
            cc["headline_vs_copy"] = out["roofline"]["achieved"] / cc["achieved"]
            out["roofline"]["copy_ceiling"] = cc
        except Exception as e:
            print(f"bench.py: copy_ceiling failed: {e}", file=sys.stderr)
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args.fft, host_cores(), args.cpu_seconds)
            except Exception as e:  # the baseline is reported, never required
                out["cpu_baseline"] = None
                print(f"bench.py: cpu_baseline failed: {e}", file=sys.stderr)

    batch.close()
    # BASELINE configs[4] in the line the driver's one command prints: at every N, all ranks (the collective is the point)
    cfg4 = None
    if not args.no_other_configs and not args.bars_allgather and world > 1:
        # A collective that never completes (a rank that died, a fabric problem) must not take the headline with it: it has been
        # measured by now.  After CFG4_WATCHDOG_S every rank gives up on its own; rank 0 prints the line first.
        import threading

        def give_up():
            if rank == 0:
                out["configs4"] = {"error": f"configs[4] region did not finish within {CFG4_WATCHDOG_S} s (collective hung?); headline unaffected"}
                print(json.dumps(out), flush=True)
            print(f"bench.py: rank {rank}: configs4 watchdog fired after {CFG4_WATCHDOG_S} s", file=sys.stderr, flush=True)
            os._exit(0)

        watchdog = threading.Timer(CFG4_WATCHDOG_S, give_up)
        watchdog.daemon = True
        watchdog.start()
        try:
            cfg4 = configs4_region(wf, torch, dist, rank, world, local_rank, min(args.steps, 300))
        except Exception as e:
            cfg4 = {"error": str(e)}
            print(f"bench.py: configs4 failed on rank {rank}: {e}", file=sys.stderr)
        # (the timer keeps running over the closing barrier below: a rank that failed above leaves the others waiting there)
    if rank == 0 and world == 1 and not args.no_other_configs and not args.bars_allgather:
        try:  # before the other shapes allocate and free their tens of gigabytes (see pcie_inclusive)
            out["pcie_inclusive"] = pcie_inclusive(wf, cfg, args.streams, local_rank)
        except Exception as e:
            print(f"bench.py: pcie_inclusive failed: {e}", file=sys.stderr)
        try:
            out["other_configs"] = other_configs(wf, local_rank, 0 if args.no_cpu_baseline else host_cores())
        except Exception as e:
            print(f"bench.py: other_configs failed: {e}", file=sys.stderr)
    if world == 1 and not args.no_other_configs and not args.bars_allgather:
        try:  # last at N=1: its 8192-stream rings are the largest allocation of the run (see pcie_inclusive)
            cfg4 = configs4_region(wf, torch, None, 0, 1, local_rank, min(args.steps, 300))
        except Exception as e:
            cfg4 = {"error": str(e)}
            print(f"bench.py: configs4 failed: {e}", file=sys.stderr)
    if rank == 0:
        if cfg4 is not None:
            out["configs4"] = cfg4
        if world == 1 and not args.no_other_configs and not args.bars_allgather:
            out["c_abi_multi"] = c_abi_multi()
        # One compact object with every figure that matters, twice: inside `roofline` (a standard key: the driver keeps those whole)
        # and as the LAST key of the line (the driver keeps the last 8 KB of stdout) -- the long per-shape objects in between may
        # be cut, the figures are not.
        def brief(r):
            rf = (r or {}).get("roofline") or {}
            return {k: (round(v, 4) if isinstance(v, float) else v) for k, v in
                    (("frac", rf.get("frac")), ("frac_events", rf.get("frac_events")), ("frac_trace", rf.get("frac_trace")),
                     ("traffic_ratio", (rf.get("traffic") / rf["algorithmic_bytes_per_tick"]) if rf.get("traffic") and rf.get("algorithmic_bytes_per_tick") else None),
                     ("ms_per_step", r.get("ms_per_step") if r else None)) if v is not None} if rf else {"error": (r or {}).get("error", "not measured")}
        shapes = {"cfg3_n4096 (headline)": {"frac": round(out["roofline"]["frac"], 4), "frac_events": round(out["roofline"]["frac_events"], 4),
                                             "frac_trace": out["roofline"].get("frac_trace"), "ms_per_step": round(out["ms_per_step"], 5)}}
        for r in out.get("other_configs") or []:
            shapes[r.get("shape") or r.get("name", "?")] = brief(r)
        if cfg4 is not None:
            shapes["configs4 (process group, gather under the next tick)"] = brief(cfg4)
        if out.get("c_abi_multi") is not None:
            shapes["c_abi_multi (wf_hip_multi_*, gather behind every tick)"] = brief(out["c_abi_multi"])
        if out.get("pcie_inclusive"):
            shapes["pcie_inclusive (host-fed headline)"] = {"spectra_per_s": round(out["pcie_inclusive"]["value"]), "host_GBps": round(out["pcie_inclusive"]["host_GBps"], 2)}
        summary = {"scaling_point": out["config"]["scaling_point"], "frac_of_8TBps_by_shape": shapes}
        out["roofline"]["summary"] = summary
        out["summary"] = summary  # (last key: survives a truncated head)
        print(json.dumps(out), flush=True)
    if dist is not None:
        if cfg4 is not None and "error" in cfg4:
            os._exit(0)  # this rank left the collective sequence of the configs[4] region: no barrier can be trusted any more
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
