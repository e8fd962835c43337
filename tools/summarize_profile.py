#!/usr/bin/env python3
"""Turns the rocprofv3 outputs of tools/profile_gpu.sh (gpurun_out/prof/<tag>/) into the small, committed
summaries under profiles/:  <name>_kernel_stats.csv  (the --kernel-trace --stats table, verbatim),
<name>_pmc.json (per-launch means of every counter + HBM bytes per launch, corrected as
MI355X_MICROARCH.md prescribes: FETCH_SIZE counts 64 B per 128 B request on gfx950 -> x2; both are in KiB)."""
import csv, glob, json, collections, shutil, sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def main(tag, name, streams=4096, fft=4096, match="spectrum_tick", command=None, kernel=None):
    src = ROOT / "gpurun_out" / "prof" / tag
    dst = ROOT / "profiles"
    dst.mkdir(exist_ok=True)
    shutil.copy(src / "stats" / "stats_kernel_stats.csv", dst / f"{name}_kernel_stats.csv")
    tot, kname, res = {}, None, {}
    per_kernel = collections.defaultdict(dict)  # kernel name -> counter -> mean per launch (paths of several kernels per tick)
    for f in sorted(glob.glob(str(src / "pmc_*" / "pmc_counter_collection.csv"))):
        agg = collections.defaultdict(list)
        by_k = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if match in r["Kernel_Name"]:
                kname = r["Kernel_Name"]
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                by_k[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
                res = {k: r[k] for k in ("VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size") if k in r}
        for k, v in agg.items():
            tot[k] = sum(v) / len(v)
        for kn, cs in by_k.items():
            for k, v in cs.items():
                per_kernel[kn][k] = sum(v) / len(v)
    stats, stats_all = {}, {}
    for r in csv.DictReader(open(src / "stats" / "stats_kernel_stats.csv")):
        if match in r["Name"]:
            stats = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"])}
            stats_all[r["Name"]] = dict(stats)
    # What the profiled command said about itself (its JSON line in stats.log): the library's kernel name -- bench.py replays a
    # summary only for the kernel it was taken on -- and the launches one tick issues.
    said = {}
    log = src / "stats.log"
    if log.exists():
        for line in log.read_text(errors="replace").splitlines():
            if line.startswith("{"):
                try:
                    d = json.loads(line)
                except Exception:
                    continue
                r = d.get("roofline") or {}
                algo = r.get("algorithmic_bytes_per_tick") or d.get("algorithmic_bytes_per_tick")
                if algo is None and d.get("GBps") and d.get("ms"):  # tools/quick_bench.py: GB/s of algorithmic bytes over ms per tick
                    algo = round(d["GBps"] * d["ms"] * 1e6)
                said = {"kernel": r.get("kernel") or d.get("kernel"), "launches_per_tick": r.get("kernel_launches_per_tick"),
                        "algorithmic_bytes_per_tick": algo}
    # The trace: which launches make a tick.  Launches of one tick may overlap (lanes: wf_hip_tick issues the batch as slices
    # on several HIP streams) or follow each other (the transforms beyond a CU's LDS: rows kernel, then epilogue), and a
    # command may run several batches one after the other, each on streams of its own (tools/wave_bench.py: three shapes).
    # So: the batch = the HIP streams whose activity overlaps the last launch's stream in time (sequential batches do not
    # overlap); a tick of it = one launch of every distinct kernel on every one of those streams; the tick span is
    # (last end - first start) / ticks over the last 40 % of the batch's ticks (steady state: every profiled command spends
    # at least half of its ticks on the lead-in its timed region follows), profiler attached.
    trace = {}
    tr = src / "stats" / "stats_kernel_trace.csv"
    if tr.exists():
        rows = sorted((r for r in csv.DictReader(open(tr)) if match in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
        if rows:
            iv = {}
            for r in rows:
                a, b = iv.get(r["Stream_Id"], (1 << 62, 0))
                iv[r["Stream_Id"]] = (min(a, int(r["Start_Timestamp"])), max(b, int(r["End_Timestamp"])))
            last = rows[-1]["Stream_Id"]
            la, lb = iv[last]
            group = {sid for sid, (a, b) in iv.items() if min(b, lb) - max(a, la) > 0.5 * min(b - a, lb - la)}
            rows = [r for r in rows if r["Stream_Id"] in group]
            lanes = len(group)
            names = sorted({r["Kernel_Name"] for r in rows})
            per_tick = lanes * len(names)
            if said.get("launches_per_tick") and len(names) == 1:  # (the same kernel twice in sequence: mono mixdown of split geometries)
                per_tick = max(per_tick, int(said["launches_per_tick"]))
            ticks = len(rows) // per_tick
            if ticks > 2:
                # tick k = launches [k * per_tick, (k + 1) * per_tick) in start order; its period = first start of tick k + 1 - first start
                # of tick k.  The figure is the MEDIAN period over the last 40 % of the ticks: (last end - first start) / ticks, the
                # round-4 form, let one host stall inside the window (the profiler's own buffer flushes) stretch every tick of it
                # (profiles/r04i_cfg3_8192streams: 141.6 us where the launches average 107.6).  Both are kept; a window whose
                # mean exceeds the median by more than 15 % says so.
                first = (ticks - max(2, ticks * 2 // 5))
                starts = [min(int(r["Start_Timestamp"]) for r in rows[k * per_tick:(k + 1) * per_tick]) for k in range(first, ticks)]
                periods = sorted(b - a for a, b in zip(starts, starts[1:]))
                median = periods[len(periods) // 2] if len(periods) % 2 else 0.5 * (periods[len(periods) // 2 - 1] + periods[len(periods) // 2])
                body = rows[first * per_tick:]
                span = max(int(r["End_Timestamp"]) for r in body) - min(int(r["Start_Timestamp"]) for r in body)
                mean = span / (len(body) // per_tick)
                trace = {"launches": len(rows), "launches_per_tick": per_tick, "concurrent_streams": lanes, "kernels_per_tick": names,
                         "ticks_in_span": len(body) // per_tick, "tick_span_ns": median, "tick_span_mean_ns": mean,
                         "stalled_window": bool(mean > 1.15 * median),
                         "note": "a tick = one launch of each of %d kernel(s) on each of %d concurrently used HIP stream(s); tick_span_ns = the median "
                                 "distance between the first launches of consecutive ticks over the last 40 %% of the ticks (steady state: behind the "
                                 "command's lead-in), profiler attached; tick_span_mean_ns = (last end - first start) / ticks of the same window "
                                 "(one host stall stretches it: stalled_window); kernel_stats averages every launch of the run, lead-in included" % (len(names), lanes)}
    fetch_b = tot.get("FETCH_SIZE", 0) * 1024 * 2
    write_b = tot.get("WRITE_SIZE", 0) * 1024
    cyc = tot.get("GRBM_GUI_ACTIVE", 0) / 8
    out = {
        "tag": tag, "command": command or "python bench.py --steps 30 --warmup 3 --no-cpu-baseline (tools/profile_gpu.sh)",
        "kernel_rocprof_name": kname, "streams": streams, "fft_size": fft,
        "kernel": said.get("kernel") or kernel or (f"spectrum_tick_kernel<N={fft},T=128,R=8x16x16,SPW=2>" if fft == 4096 else None),
        "algorithmic_bytes_per_tick": said.get("algorithmic_bytes_per_tick"),
        "kernel_stats": stats, "trace": trace, "dispatch": res,
        "hbm_bytes_per_launch": fetch_b + write_b,
        "hbm_read_bytes_per_launch": fetch_b, "hbm_write_bytes_per_launch": write_b,
        "hbm_bytes_per_tick": (fetch_b + write_b) * (trace.get("concurrent_streams", 1) if trace else 1),
        # a tick of several different kernels in sequence (the transforms beyond a CU's LDS): per kernel, and their sum
        "kernels": {kn: {"avg_ns": stats_all.get(kn, {}).get("avg_ns"), "calls": stats_all.get(kn, {}).get("calls"),
                         "hbm_bytes_per_launch": cs.get("FETCH_SIZE", 0) * 2048 + cs.get("WRITE_SIZE", 0) * 1024} for kn, cs in per_kernel.items()} if len(per_kernel) > 1 else None,
        # (x the lanes: with the batch issued as slices on several streams a launch covers one slice -- round 5 put the row paths of the
        # sizes above 16384 on two lanes, and the sum of one launch of each kernel then read half a tick)
        "hbm_bytes_per_tick_all_kernels": (trace.get("concurrent_streams", 1) if trace else 1) * sum(cs.get("FETCH_SIZE", 0) * 2048 + cs.get("WRITE_SIZE", 0) * 1024 for cs in per_kernel.values()) if len(per_kernel) > 1 else None,
        "ns_per_tick_all_kernels": sum((stats_all.get(kn, {}).get("avg_ns") or 0) for kn in per_kernel) if len(per_kernel) > 1 else None,
        "correction": "FETCH_SIZE (KiB) x2 per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE (KiB) as reported",
        "counters_mean_per_launch": tot,
        "derived": {
            "valu_insts_per_wave": tot.get("SQ_INSTS_VALU", 0) / max(tot.get("SQ_WAVES", 1), 1),
            "valu_busy_frac": tot.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (1024 * cyc) if cyc else None,
            "avg_waves_per_simd": tot.get("SQ_WAVE_CYCLES", 0) * 4 / (1024 * cyc) if cyc else None,
            "wait_any_frac": tot.get("SQ_WAIT_ANY", 0) / max(tot.get("SQ_WAVE_CYCLES", 1), 1),
            "lds_bank_conflict_frac": tot.get("SQ_LDS_BANK_CONFLICT", 0) / max(tot.get("SQ_LDS_IDX_ACTIVE", 1), 1),
            "kernel_cycles": cyc,
        },
    }
    (dst / f"{name}_pmc.json").write_text(json.dumps(out, indent=1))
    print(json.dumps({k: out[k] for k in ("kernel_stats", "trace", "hbm_bytes_per_launch", "hbm_bytes_per_tick", "derived")}, indent=1))


if __name__ == "__main__":
    # usage: summarize_profile.py TAG NAME [kernel-name-substring [streams [fft [command]]]]
    a = sys.argv
    main(a[1], a[2], streams=int(a[4]) if len(a) > 4 else 4096, fft=int(a[5]) if len(a) > 5 else 4096,
         match=a[3] if len(a) > 3 else "spectrum_tick", command=a[6] if len(a) > 6 else None,
         kernel=a[3] if len(a) > 3 and a[3] != "spectrum_tick" else None)
