#!/usr/bin/env python3
"""tools/node_check.py -- first thing to run on a multi-GPU node nobody could rehearse on: enumerates the
devices and the links between them, runs BASELINE configs[4]'s shape through the C ABI's multi-device group (wf_hip_multi_*) over
every device, one leg per process -- the default transport (ncclAllGather of the dlopen()ed librccl.so where the devices are
distinct); RCCL with its channel count capped at one and at two (how many CUs the collective's kernels take from a tick that
fills them in whole rounds); RCCL with the gather stream on a hardware queue of its own; the peer transport with the tick
kernels storing their slice into every device's result (no kernel, no copy in the exchange) and with copies behind the tick --
and prints ONE JSON object: per-device tick times with and without the gather and their difference per leg, the gather's own
time, the link type / hop count of every device pair, and whether every device's gathered copy equals the shards' own bars.  Exit code 0 only if everything verified; otherwise one line on stderr says which transport and which
device failed.

    python tools/node_check.py [--devices N] [--streams-per-device 8192] [--ticks 200]
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
HOP, FFT, SEED = 800, 4096, 0x5741564546524D31
LINK = {0: "hypertransport", 1: "qpi", 2: "pcie", 3: "infiniband", 4: "xgmi"}


def links(n):
    """hipExtGetLinkTypeAndHopCount for every ordered pair of devices"""
    out = []
    try:
        hip = C.CDLL("libamdhip64.so")
        f = hip.hipExtGetLinkTypeAndHopCount
        f.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        for a in range(n):
            for b in range(n):
                if a == b:
                    continue
                t, h = C.c_uint32(0), C.c_uint32(0)
                rc = f(a, b, C.byref(t), C.byref(h))
                out.append({"from": a, "to": b, "type": LINK.get(int(t.value), str(int(t.value))) if rc == 0 else f"error {rc}", "hops": int(h.value)})
    except Exception as e:  # reported, not fatal: the gather below is the real test
        out.append({"error": str(e)})
    return out


def run(wf, np, devices, transport, streams_per_device, ticks):
    old = os.environ.pop("WF_HIP_MULTI_TRANSPORT", None)
    if transport:
        os.environ["WF_HIP_MULTI_TRANSPORT"] = transport
    try:
        cfg = wf.Config.defaults(fft_size=FFT, stereo=1, slope=1.0, tsmoothing=wf.TSMOOTH["exponential"], gravity=0.65, bars=1,
                                 interp_mode=wf.INTERP["lanczos"])
        flags, depth = wf.TICK_NO_DECIBELS, 16
        total = streams_per_device * len(devices)
        with wf.MultiBatch(cfg, total, devices, ring_frames=FFT + HOP * (depth + 1)) as m:
            m.push_synth(SEED, 0, HOP * depth)
            m.sync()
            first = HOP * (depth - 1)
            probe, _ = m.time_ticks(8, HOP, first, gather=True, flags=flags)
            m.time_ticks(int(40.0 / max(probe, 1e-4)) + 1, HOP, first, gather=True, flags=flags)       # clocks settle
            ms_plain, per_plain = m.time_ticks(ticks, HOP, first, gather=False, flags=flags)
            ms_gather, per_gather = m.time_ticks(ticks, HOP, first, gather=True, flags=flags)
            m.sync()
            t0 = time.perf_counter()
            for _ in range(50):                      # the exchange alone: 50 gathers of the same bars, then one wait
                m.allgather_bars()
            m.sync()
            gather_us = (time.perf_counter() - t0) / 50 * 1e6
            own = m.bars()
            bad = [i for i in range(m.n_devices) if not np.array_equal(m.gathered(i), own)]
            return {"transport_asked": transport or "default", "transport": m.transport, "devices": devices, "streams_total": total,
                    "ms_per_tick_without_gather": {"max": ms_plain, "per_device": per_plain},
                    "ms_per_tick_with_gather": {"max": ms_gather, "per_device": per_gather},
                    "gather_alone_us": gather_us, "gathered_bytes_per_device": int(own.nbytes),
                    "finite": bool(np.isfinite(own).all()), "devices_with_a_wrong_copy": bad, "verified": not bad and bool(np.isfinite(own).all())}
    except Exception as e:
        return {"transport_asked": transport or "default", "devices": devices, "error": str(e), "verified": False}
    finally:
        os.environ.pop("WF_HIP_MULTI_TRANSPORT", None)
        if old is not None:
            os.environ["WF_HIP_MULTI_TRANSPORT"] = old


LEGS = [
    # (name, transport asked for, extra environment): every leg in a process of its own -- RCCL reads its environment once
    ("default", None, {}),
    ("rccl, one channel", "rccl", {"NCCL_MAX_NCHANNELS": "1"}),
    ("rccl, two channels", "rccl", {"NCCL_MAX_NCHANNELS": "2"}),
    ("rccl, gather stream on a hardware queue of its own", "rccl", {"WF_HIP_MULTI_GATHER_PRIORITY": "high"}),
    ("peer, the tick kernels store into every device's result", "peer", {}),
    ("peer, copies behind the tick", "peer", {"WF_HIP_MULTI_MIRROR": "send"}),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--devices", type=int, default=0, help="0: every visible device")
    ap.add_argument("--streams-per-device", type=int, default=8192)
    ap.add_argument("--ticks", type=int, default=200)
    ap.add_argument("--leg", type=int, default=-1, help="(internal) run one leg of LEGS in this process and print its object")
    args = ap.parse_args()
    import numpy as np
    import waveform_amd as wf
    have = wf.device_count()
    n = have if args.devices <= 0 else min(args.devices, have)
    if n < 1:
        print("node_check: no usable gfx950 device", file=sys.stderr)
        return 3
    devices = list(range(n))
    if args.leg >= 0:
        name, transport, _ = LEGS[args.leg]
        if n == 1 and transport == "peer":  # one device: the peer path with two shards on it (the shard arithmetic, the threads, the double buffering)
            r = run(wf, np, [0, 0], "peer", args.streams_per_device // 2, args.ticks)
        else:
            r = run(wf, np, devices, transport, args.streams_per_device, args.ticks)
        r["leg"] = name
        print(json.dumps(r), flush=True)
        return 0
    import subprocess
    runs = []
    for i, (name, transport, env) in enumerate(LEGS):
        cmd = [sys.executable, os.path.abspath(__file__), "--leg", str(i), "--devices", str(n), "--streams-per-device", str(args.streams_per_device),
               "--ticks", str(args.ticks)]
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
            lines = [l for l in p.stdout.strip().splitlines() if l.startswith("{")]
            r = json.loads(lines[-1]) if lines else {"leg": name, "error": f"rc {p.returncode}: {p.stderr.strip()[-300:]}", "verified": False}
        except subprocess.TimeoutExpired:
            r = {"leg": name, "error": "timed out after 300 s (a collective nobody answered?)", "verified": False}
        r["environment"] = env
        if "ms_per_tick_with_gather" in r and "ms_per_tick_without_gather" in r:
            r["gather_costs_per_tick_us"] = round((r["ms_per_tick_with_gather"]["max"] - r["ms_per_tick_without_gather"]["max"]) * 1e3, 2)
        runs.append(r)
    out = {"devices_visible": have, "devices_used": n, "links": links(n), "runs": runs}
    print(json.dumps(out), flush=True)
    rc = 0
    for r in runs:
        if not r.get("verified"):
            rc = 1
            print(f"node_check: leg '{r.get('leg')}' (transport {r.get('transport', r.get('transport_asked'))}) over devices {r.get('devices')}: "
                  + (r.get("error") or f"device indices {r.get('devices_with_a_wrong_copy')} hold a gathered copy that differs from the shards' bars"), file=sys.stderr)
    return rc


if __name__ == "__main__":
    sys.exit(main())
