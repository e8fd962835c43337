#!/usr/bin/env python3
"""BASELINE configs[4] through the C ABI's single-process multi-device group (wf_hip_multi_*, include/wf_hip.h): 8192 stereo
streams per visible device (65536 on a full node), FFT 4096, EMA + slope, 26 Lanczos bars per channel, bars-only ticks, and the
all-gather of the bar heights behind every tick -- one host thread per device, ncclAllGather of the dlopen()ed librccl.so on
the devices' gather streams (peer copies where RCCL is not available).  Prints one JSON object.  No torch in this process:
the C++ host path is the thing measured.

    python tools/multi_bench.py [--devices N] [--streams-per-device 8192] [--ticks 300]
"""
from __future__ import annotations

import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

HOP, FFT, SEED = 800, 4096, 0x5741564546524D31
HBM_PEAK_GBPS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--devices", type=int, default=0, help="0: every visible device")
    ap.add_argument("--streams-per-device", type=int, default=8192)
    ap.add_argument("--ticks", type=int, default=300)
    ap.add_argument("--lead-in-ms", type=float, default=40.0)
    args = ap.parse_args()

    import numpy as np
    import waveform_amd as wf

    have = wf.device_count()
    n = have if args.devices <= 0 else min(args.devices, have)
    if n < 1:
        print(json.dumps({"error": "no device"}))
        return 3
    cfg = wf.Config.defaults(fft_size=FFT, stereo=1, slope=1.0, window=wf.WINDOW["hann"], tsmoothing=wf.TSMOOTH["exponential"],
                             gravity=0.65, bars=1, interp_mode=wf.INTERP["lanczos"])
    flags = wf.TICK_NO_DECIBELS
    total, depth = args.streams_per_device * n, 16
    with wf.MultiBatch(cfg, total, list(range(n)), ring_frames=FFT + HOP * (depth + 1)) as m:
        m.push_synth(SEED, 0, HOP * depth)
        m.sync()
        first = HOP * (depth - 1)
        probe, _ = m.time_ticks(8, HOP, first, gather=True, flags=flags)
        warm = int(args.lead_in_ms / max(probe, 1e-4)) + 1
        m.time_ticks(warm, HOP, first, gather=True, flags=flags)
        t0 = time.perf_counter()
        ms, per = m.time_ticks(args.ticks, HOP, first, gather=True, flags=flags)
        m.sync()
        wall = time.perf_counter() - t0
        own = m.bars()
        ok = bool(np.isfinite(own).all())
        for i in range(n):
            ok = ok and bool(np.array_equal(m.gathered(i), own))
        algo = m.algorithmic_bytes_per_tick(flags)
        spectra = total * m.capture_channels
        out = {
            "name": f"BASELINE configs[4] through wf_hip_multi_* (single process, one host thread per device): {total} stereo streams over "
                    f"{n} device(s), FFT {FFT}, EMA + slope, 26 Lanczos bars per channel, bars-only ticks, all-gather of the bars behind every tick",
            "n_devices": n, "transport": m.transport, "streams_total": total, "spectra_per_tick": spectra, "ticks": args.ticks,
            "warmup": warm + 8, "value": spectra * args.ticks / wall, "unit": "spectra/s", "ms_per_step": wall * 1e3 / args.ticks,
            "device_ms_per_tick": {"min": min(per), "max": max(per), "per_device": per},
            "gathered_bytes_per_device_per_tick": int(total * m.display_channels * m.num_bars * 4),
            "verified": ok,
            "verification": "every device's gathered copy == the bars read back shard by shard (bit for bit); all values finite",
            # per device: the slowest device's time against one device's share of the bytes
            "roofline": {"bound": "hbm", "achieved": algo / n / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": algo / n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "per": "device (slowest)"},
        }
    print(json.dumps(out), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
