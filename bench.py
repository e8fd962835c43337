#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X spectrum path (contract: see the task brief).

Metric (BASELINE.json): spectra/sec at FFT=4096, batch=4096 streams, with the achieved HBM
GB/s of the fused kernel against the chip's peak.

Workload at every N: BASELINE.json configs[2] per GPU -- 4096 independent stereo streams
(8192 spectra per tick), FFT 4096, Hann window, EXPONENTIAL smoothing g=0.65, slope 1.0,
48 kHz synthetic white noise (include/wf_synth.h), hop 800 samples (60 fps).  Weak scaling:
every rank owns its own 4096 streams, no data-path collective (the streams share nothing,
SURVEY.md §8(e)).

A "step" is one tick_spectrum over the whole batch = one launch of the fused kernel.  All
audio for warm-up + timed ticks is generated into the device rings before the timed region
(inputs resident in HBM); tick i analyses the window that ends (steps-1-i)*hop frames before
the newest sample, i.e. exactly the ring contents the reference would see at that video
frame (its own A/V-sync path, src/source_generic.cpp:50-59).

One JSON line on rank 0.  Extra objects:
  roofline     algorithmic bytes per launch / average kernel duration (HIP events on the
               library's stream, same timed region) vs 8 TB/s HBM peak
  cpu_baseline the reference's own AVX2 path (oracle/_ref/libwfref.so: verbatim TUs + vendored
               FFTW) timed on this box's host cores on a bounded sample (rank 0, N=1 only)
"""
from __future__ import annotations

import argparse
import json
import os
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


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--streams", type=int, default=STREAMS_PER_GPU, help="streams per GPU")
    ap.add_argument("--fft", type=int, default=FFT_SIZE)
    ap.add_argument("--bars-allgather", action="store_true",
                    help="BASELINE configs[4] shape: bars (26 Lanczos bars per channel) computed in the tick and all-gathered "
                         "across ranks (RCCL over xGMI) after every step; changes the workload, so it is off by default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=30.0, help="CPU work budget (core-seconds) of the baseline leg")
    return ap.parse_args()


def cpu_baseline(fft: int, cores: int, budget_core_s: float):
    """The reference's AVX2+FMA3 path on the host cores, bounded sample."""
    from oracle import wfref
    if not wfref.available():
        return None
    settings = dict(fft_size=fft, enable_large_fft=True, channel_mode="stereo", slope=1.0, window="hann",
                    temporal_smoothing="exp_moving_avg", gravity=0.65)
    # calibrate on one core, then size the sample to the budget
    v1, _ = wfref.bench("avx2", settings, 4, 1, 8, 64, hop=HOP, seed=SEED)
    if v1 <= 0:
        return None
    per_core = max(budget_core_s / max(cores, 1), 0.25)          # seconds of work per thread
    streams_per_thread = 8
    ticks = int(max(64, min(16384, per_core * v1 / (2 * streams_per_thread))))
    n_streams = streams_per_thread * cores
    v, el = wfref.bench("avx2", settings, n_streams, cores, 16, ticks, hop=HOP, seed=SEED)
    return {
        "value": v, "unit": "spectra/s", "cores": cores, "kind": "reference",
        "sample": f"WAVSourceAVX2::tick_spectrum (verbatim reference + vendored FFTW 3.3.11), {n_streams} stereo streams x "
                  f"{ticks} ticks, hop {HOP}, FFT {fft}, {cores} threads, {el:.2f} s wall; 1 thread: {v1:.0f} spectra/s",
    }


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


def pmc_traffic(kernel: str, streams: int):
    """HBM bytes per launch from committed rocprofv3 PMC passes (profiles/*_pmc.json), or None."""
    best = None
    for p in sorted((ROOT / "profiles").glob("*_pmc.json")):
        try:
            d = json.loads(p.read_text())
        except Exception:
            continue
        if d.get("kernel") == kernel and d.get("streams") == streams and d.get("hbm_bytes_per_launch"):
            best = d["hbm_bytes_per_launch"]
    return best


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus}", file=sys.stderr)
            sys.exit(2)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import torch
    import waveform_amd as wf

    if not torch.cuda.is_available():
        print("bench.py: no GPU visible (torch.cuda.is_available() is False)", file=sys.stderr)
        sys.exit(3)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    cfg = wf.Config.defaults(fft_size=args.fft, stereo=1, slope=1.0, window=wf.WINDOW["hann"],
                             tsmoothing=wf.TSMOOTH["exponential"], gravity=0.65)
    if args.bars_allgather:
        cfg.bars = 1
        cfg.interp_mode = wf.INTERP["lanczos"]
    total_ticks = args.warmup + args.steps
    # audio of up to MAX_DEPTH consecutive ticks is resident; longer runs walk the same windows again (same work per step)
    MAX_DEPTH = 512
    depth = min(total_ticks, MAX_DEPTH)
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
        from waveform_amd.dist import shard_streams, allgather_bars
        shard = shard_streams(args.streams * world, rank, world)
        local_bars = torch.empty((args.streams, batch.display_channels, batch.num_bars), dtype=torch.float32, device="cuda")

        def gather():
            batch.copy_bars_to_device(local_bars.data_ptr())       # D2D on the library's stream (synchronised)
            return allgather_bars(local_bars, shard)               # one all_gather_into_tensor

    def run(n_ticks):
        """n_ticks steps over the resident audio, oldest window first; returns the average fused-kernel duration in ms"""
        ms, done = 0.0, 0
        while done < n_ticks:
            n = min(depth, n_ticks - done)
            if gather is None:
                ms += batch.time_ticks(n, HOP, HOP * (n - 1)) * n
            else:
                for i in range(n):
                    ms += batch.time_ticks(1, HOP, HOP * (n - 1 - i))
                    gather()
            done += n
        return ms / n_ticks

    # warm-up: W untimed steps
    if args.warmup > 0:
        run(args.warmup)
    barrier()
    t0 = time.perf_counter()
    # K timed steps: K launches of the fused kernel, HIP events around them on the library's stream
    kernel_ms = run(args.steps)
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        k = torch.tensor([kernel_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(k, op=dist.ReduceOp.MAX)
        kernel_ms = float(k.item())

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = spectra_per_step * world * args.steps / elapsed
        algo_bytes = batch.algorithmic_bytes_per_tick()
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "spectra/sec at FFT=4096, batch=4096 streams; achieved HBM GB/s vs peak",
            "value": value,
            "unit": "spectra/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": (f"BASELINE configs[{4 if args.bars_allgather else 2}]{' shape (per GPU)' if args.bars_allgather else ''}: "
                             f"{args.streams} independent stereo streams per GPU ({spectra_per_step} spectra/tick), "
                             f"FFT={args.fft}, Hann, EMA g=0.65 + slope 1.0"
                             f"{', 26 Lanczos bars per channel all-gathered' if args.bars_allgather else ''}, "
                             f"48 kHz counter-hash white noise, hop {HOP}"),
                "streams_per_gpu": args.streams, "fft_size": args.fft, "hop": HOP,
                "parallelism": (f"streams sharded over {world} GPU(s); bar heights all-gathered after every step" if args.bars_allgather
                                else f"streams sharded over {world} GPU(s), no data-path collective"),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                "traffic": pmc_traffic(batch.kernel_name(), args.streams),
                "kernel": batch.kernel_name(), "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": algo_bytes,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args.fft, host_cores(), args.cpu_seconds)
            except Exception as e:  # the baseline is reported, never required
                out["cpu_baseline"] = None
                print(f"bench.py: cpu_baseline failed: {e}", file=sys.stderr)
        print(json.dumps(out), flush=True)

    batch.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
