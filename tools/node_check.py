#!/usr/bin/env python3
"""tools/node_check.py -- first thing to run on a multi-GPU node nobody could rehearse on: enumerates the
devices and the links between them, runs BASELINE configs[4]'s shape through the C ABI's multi-device group (wf_hip_multi_*) over
every device, one leg per process -- the default transport (ncclAllGather of the dlopen()ed librccl.so where the devices are
distinct); RCCL with its channel count capped at one and at two (how many CUs the collective's kernels take from a tick that
fills them in whole rounds); RCCL with the gather stream on a hardware queue of its own; the default transport with the tick kernels
writing the send buffers (no copy behind the tick: equal or slower on one device, for a node to decide); the peer transport with the tick
kernels storing their slice into every device's result (no kernel, no copy in the exchange) and with copies behind the tick --
and prints ONE JSON object: per-device tick times with and without the gather and their difference per leg, the gather's own
time, the link type / hop count and the peer-access answer of every device pair, and whether every device's gathered copy equals
the shards' own bars.

Exit code 0 only if everything verified; otherwise ONE LINE PER REASON on stderr:
  * RCCL refuses the device list (the default leg ran on another transport than "rccl" although the devices are distinct, or a leg
    that asked for RCCL by name failed): the library's own text says why (ncclCommInitAll's error, librccl.so not loadable);
  * peer access denied on a pair of devices (hipDeviceCanAccessPeer / hipDeviceEnablePeerAccess): the pair is named -- the tick
    kernels' direct stores into the other devices' results are then off and the bars travel by hipMemcpyPeerAsync;
  * a device more than 10 % slower than the best one on the same shard size (per-device tick times without the gather): that
    device is named with both figures -- a throttling or mis-seated part would otherwise hide in the max-over-ranks;
  * a gathered copy that differs from the shards' own bars, a leg that timed out or raised.

--host-fed adds the leg an operator sizes a live deployment by: the headline batch on EVERY device at once, each fed through its
own PCIe link every step (page-locked buffers, wf_hip_push_audio_async under the previous tick, one host thread per device), and
prints GB/s and spectra/s per device plus the node total -- one MI355X alone holds 54 GB/s (bench.py pcie_inclusive); whether
eight links hold that together depends on the host's PCIe topology, which nobody has measured.

    python tools/node_check.py [--devices N] [--streams-per-device 8192] [--ticks 200] [--host-fed] [--host-fed-only]
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


def peer_matrix(n):
    """hipDeviceCanAccessPeer and (in this process, which creates nothing else) hipDeviceEnablePeerAccess for every ordered pair"""
    out = []
    try:
        hip = C.CDLL("libamdhip64.so")
        for a in range(n):
            if hip.hipSetDevice(a) != 0:
                out.append({"from": a, "error": "hipSetDevice failed"})
                continue
            for b in range(n):
                if a == b:
                    continue
                can = C.c_int(0)
                rc = hip.hipDeviceCanAccessPeer(C.byref(can), a, b)
                en = hip.hipDeviceEnablePeerAccess(b, 0) if (rc == 0 and can.value) else -1
                ok = rc == 0 and bool(can.value) and en in (0, 704)  # hipSuccess / hipErrorPeerAccessAlreadyEnabled
                out.append({"from": a, "to": b, "can_access": bool(can.value) if rc == 0 else None, "enable_rc": en, "ok": ok})
    except Exception as e:
        out.append({"error": str(e)})
    return out


def host_fed_device(wf, np, device, streams, steps, warm, start, out):
    """bench.py's pcie_inclusive on one device, started together with the other devices' threads"""
    try:
        from tools import synth
        cfg = wf.Config.defaults(fft_size=FFT, stereo=1, slope=1.0, tsmoothing=wf.TSMOOTH["exponential"], gravity=0.65)
        packet = np.ascontiguousarray(np.broadcast_to(synth.block(SEED, 0, 1, 2, 0, HOP), (streams, 2, HOP)), np.float32)
        with wf.SpectrumBatch(cfg, streams, device=device) as b:
            pin = [wf.PinnedBuffer(packet.shape), wf.PinnedBuffer(packet.shape)]
            for p in pin:
                p.array[...] = packet
            t0 = 0.0
            for i in range(warm + steps):
                if i == warm:
                    b.sync()
                    start.wait()       # every device's timed region starts together
                    t0 = time.perf_counter()
                slot = i & 1
                b.ingest_done(slot)
                b.push_audio_async(pin[slot], streams, HOP, slot)
                b.tick()
            b.sync()
            dt = (time.perf_counter() - t0) / steps
            for p in pin:
                p.close()
        out[device] = {"device": device, "host_GBps": packet.nbytes / dt / 1e9, "spectra_per_s": streams * 2 / dt, "ms_per_step": dt * 1e3}
    except Exception as e:
        out[device] = {"device": device, "error": str(e)}
        try:
            start.abort()
        except Exception:
            pass


def host_fed(wf, np, devices, streams=4096, steps=200, warm=400):
    import threading
    out = {}
    start = threading.Barrier(len(devices))
    th = [threading.Thread(target=host_fed_device, args=(wf, np, d, streams, steps, warm, start, out)) for d in devices]
    for t in th:
        t.start()
    for t in th:
        t.join()
    per = [out.get(d, {"device": d, "error": "no result"}) for d in devices]
    good = [r for r in per if "error" not in r]
    return {"what": "the headline batch (4096 stereo streams, FFT 4096) on every device at once, one 800-frame hop per stream and step through each device's own PCIe link "
                    "(page-locked buffer -> wf_hip_push_audio_async under the previous tick -> ring append -> tick), one host thread per device",
            "streams_per_device": streams, "steps": steps, "per_device": per,
            "node_host_GBps": sum(r["host_GBps"] for r in good), "node_spectra_per_s": sum(r["spectra_per_s"] for r in good),
            "verified": len(good) == len(devices)}


def reasons(out, n):
    """one line per reason the node is not ready (see the module docstring); empty: ready"""
    why = []
    for pr in out.get("peer_access", []):
        if "error" in pr:
            why.append(f"peer access could not be queried: {pr['error']}")
        elif not pr.get("ok"):
            why.append(f"peer access denied: device {pr['from']} cannot address device {pr['to']} (hipDeviceCanAccessPeer {pr.get('can_access')}, "
                       f"hipDeviceEnablePeerAccess rc {pr.get('enable_rc')}): direct peer stores are off, the bars travel by hipMemcpyPeerAsync")
    for r in out.get("runs", []):
        leg, asked = r.get("leg"), r.get("transport_asked")
        if r.get("error"):
            kind = "RCCL refuses the device list" if (asked == "rccl" or (asked == "default" and n > 1)) and ("nccl" in r["error"].lower() or "rccl" in r["error"].lower()) else "failed"
            why.append(f"leg '{leg}' over devices {r.get('devices')}: {kind}: {r['error']}")
            continue
        if n > 1 and asked == "default" and r.get("transport") != "rccl":
            why.append(f"RCCL refuses the device list {r.get('devices')}: the default leg ran on transport '{r.get('transport')}' ({r.get('transport_note') or 'no reason given'})")
        if r.get("devices_with_a_wrong_copy"):
            why.append(f"leg '{leg}' (transport {r.get('transport')}): device indices {r['devices_with_a_wrong_copy']} hold a gathered copy that differs from the shards' bars")
        elif not r.get("verified"):
            why.append(f"leg '{leg}' (transport {r.get('transport')}): not verified (non-finite bars?)")
        per = (r.get("ms_per_tick_without_gather") or {}).get("per_device") or []
        if len(per) > 1 and min(per) > 0 and len(set(r.get("devices", []))) == len(per):
            best = min(per)
            for i, ms in enumerate(per):
                if ms > 1.10 * best:
                    why.append(f"leg '{leg}': device {r['devices'][i]} takes {ms * 1e3:.1f} us per tick, {100 * (ms / best - 1):.0f} % more than the best device "
                               f"({best * 1e3:.1f} us) on the same shard size")
    hf = out.get("host_fed")
    if hf is not None and not hf.get("verified"):
        for r in hf.get("per_device", []):
            if "error" in r:
                why.append(f"host-fed leg: device {r['device']}: {r['error']}")
    return why


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
            return {"transport_asked": transport or "default", "transport": m.transport, "transport_note": m.transport_note, "devices": devices, "streams_total": total,
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
    ("default transport, the tick kernels write the send buffers themselves", None, {"WF_HIP_MULTI_MIRROR": "1"}),
    ("peer, the tick kernels store into every device's result", "peer", {}),
    ("peer, copies behind the tick", "peer", {"WF_HIP_MULTI_MIRROR": "send"}),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--devices", type=int, default=0, help="0: every visible device")
    ap.add_argument("--streams-per-device", type=int, default=8192)
    ap.add_argument("--ticks", type=int, default=200)
    ap.add_argument("--leg", type=int, default=-1, help="(internal) run one leg of LEGS in this process and print its object")
    ap.add_argument("--host-fed", action="store_true", help="also: the headline batch on every device at once, fed through PCIe every step; GB/s per device")
    ap.add_argument("--host-fed-only", action="store_true", help="only that leg")
    ap.add_argument("--host-fed-streams", type=int, default=4096)
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
    for i, (name, transport, env) in enumerate([] if args.host_fed_only else LEGS):
        cmd = [sys.executable, os.path.abspath(__file__), "--leg", str(i), "--devices", str(n), "--streams-per-device", str(args.streams_per_device),
               "--ticks", str(args.ticks)]
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
            lines = [l for l in p.stdout.strip().splitlines() if l.startswith("{")]
            r = json.loads(lines[-1]) if lines else {"leg": name, "transport_asked": transport or "default", "devices": devices,
                                                     "error": f"rc {p.returncode}: {p.stderr.strip()[-300:]}", "verified": False}
        except subprocess.TimeoutExpired:
            r = {"leg": name, "transport_asked": transport or "default", "devices": devices, "error": "timed out after 300 s (a collective nobody answered?)", "verified": False}
        r["environment"] = env
        if "ms_per_tick_with_gather" in r and "ms_per_tick_without_gather" in r:
            r["gather_costs_per_tick_us"] = round((r["ms_per_tick_with_gather"]["max"] - r["ms_per_tick_without_gather"]["max"]) * 1e3, 2)
        runs.append(r)
    out = {"devices_visible": have, "devices_used": n, "links": links(n), "peer_access": peer_matrix(n), "runs": runs}
    if args.host_fed or args.host_fed_only:
        out["host_fed"] = host_fed(wf, np, devices, streams=args.host_fed_streams)
    why = reasons(out, n)
    out["ready"] = not why
    out["reasons"] = why
    print(json.dumps(out), flush=True)
    for line in why:
        print("node_check: " + line, file=sys.stderr)
    return 1 if why else 0


if __name__ == "__main__":
    sys.exit(main())
