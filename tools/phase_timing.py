"""development aid: per-phase s_memtime stamps of the tick kernel (needs the -DWF_PHASE_TIMING build)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("WF_HIP_LIB", os.path.abspath("variants/lib_timing.so"))
import waveform_amd as wf
from tools import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
streams = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
hop, ticks = 800, 6
extra = dict(bars=1, interp_mode=1) if os.environ.get("WF_BENCH_BARS") else {}
if os.environ.get("WF_BENCH_CURVE"):
    extra = dict(curve=1, interp_mode=int(os.environ["WF_BENCH_CURVE"]))
cfg = wf.Config.defaults(fft_size=n, stereo=1, slope=1.0, **extra)
b = wf.SpectrumBatch(cfg, streams, ring_frames=n + hop * (ticks + 2))
b.push_synth(synth.DEFAULT_SEED, 0, hop * (ticks + 1))
FLAGS = int(os.environ.get("WF_BENCH_FLAGS", "0"))  # 1: WF_HIP_TICK_NO_DECIBELS (bars-only ticks)
b.time_ticks(int(os.environ.get("WF_WARM_TICKS", "600")), hop, hop * (ticks - 1), flags=FLAGS)  # settled clocks (profiles/r02j_warmup.txt)
for i in range(ticks):
    b.tick(delay_frames=hop * (ticks - 1 - i), flags=FLAGS)
b.sync()
nblk = streams * 2 // 2
buf = np.zeros(nblk * 16, np.uint64)
L = wf.lib()
L.wf_hip_debug_phase_clock.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
assert L.wf_hip_debug_phase_clock(b.h, buf.ctypes.data, buf.size) == 0
s = buf.reshape(nblk, 16).astype(np.int64)
if os.environ.get("WF_PHASE_MR"):  # the mixed-radix / Bluestein paths stamp 0, 1, 2, 8, 9, 10 only
    tot = s[:, 10] - s[:, 0]
    print("block lifetime: mean %.0f  p10 %.0f  p90 %.0f ticks of s_memtime (100 MHz)" % (tot.mean(), np.percentile(tot, 10), np.percentile(tot, 90)))
    for nm, a0, a1 in (("fetch+nz", 0, 1), ("facts xchg", 1, 2), ("transform", 2, 8), ("p4 split+smooth", 8, 9), ("dB+store", 9, 10)):
        dd = s[:, a1] - s[:, a0]
        print(f"{nm:18s} mean {dd.mean():9.1f}  ({100 * dd.mean() / tot.mean():5.1f} %)   p90 {np.percentile(dd, 90):9.1f}")
    sys.exit(0)
d = np.diff(s[:, :11], axis=1)
names = ["fetch+nz", "facts xchg", "p1 win+pass1", "sync+p2 read", "p2 dft+write", "sync+p3 read", "p3 dft+write", "sync", "p4 split+smooth", "dB+store"]
if os.environ.get("WF_HIP_KERNEL") == "pipe":
    names = ["tables+wait window", "samples+facts xchg", "p1 win+pass1", "sync+p2 read", "p2 dft+write", "sync+p3 read", "p3 dft+write+sync",
             "p4 loads+sync+dma issue", "p4 math", "dB+store"]
print("stamps are s_memtime ticks (100 MHz constant clock?) -- relative shares matter")
tot = (s[:, 10] - s[:, 0])
print("block lifetime: mean %.0f  p10 %.0f  p90 %.0f" % (tot.mean(), np.percentile(tot, 10), np.percentile(tot, 90)))
for i, nm in enumerate(names):
    print(f"{nm:18s} mean {d[:, i].mean():9.1f}  ({100 * d[:, i].mean() / tot.mean():5.1f} %)   p90 {np.percentile(d[:, i], 90):9.1f}")

print("after stamp 10 (flags store, bars if any) until the end of the kernel: mean %.0f ticks" % (s[:, 13] - s[:, 10]).mean())
if os.environ.get("WF_BENCH_CURVE"):
    print("curve: row->LDS+syncs %.0f | points + mapping + stores %.0f" % ((s[:, 12] - s[:, 10]).mean(), (s[:, 13] - s[:, 12]).mean()))
if os.environ.get("WF_BENCH_BARS") and os.environ.get("WF_HIP_BAR_PS", "1") != "0":
    print("bars (prefix-sum layout, first wavefront of the workgroup): wait for the last reads %.0f | park row + group sums %.0f | count in + wait for everybody %.0f | prefix, sub-bands, stores %.0f" %
          ((s[:, 12] - s[:, 10]).mean(), (s[:, 14] - s[:, 12]).mean(), (s[:, 15] - s[:, 14]).mean(), (s[:, 13] - s[:, 15]).mean()))
elif os.environ.get("WF_BENCH_BARS"):
    print("bars: row->LDS+syncs %.0f | A products+sync %.0f | B1 segment sums+sync %.0f | B2 bar sums+stores %.0f" %
          ((s[:, 12] - s[:, 10]).mean(), (s[:, 14] - s[:, 12]).mean(), (s[:, 15] - s[:, 14]).mean(), (s[:, 13] - s[:, 15]).mean()))
# ---- where and when: per-CU residency from HW_ID (slot 11) / XCC_ID (slot 12) ------------------------------
hw, xcc = s[:, 11], 0 * s[:, 11]
cu = ((xcc << 12) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF))
# the cycle counters of different CUs are not synchronised: normalise per CU (the launch ramp is not visible this way)
ids = np.unique(cu)
start, end = np.zeros(nblk), np.zeros(nblk)
for i in ids:
    m = cu == i
    start[m], end[m] = s[m, 0] - s[m, 0].min(), s[m, 10] - s[m, 0].min()
span = np.median([end[cu == i].max() for i in ids])
print(f"CUs seen: {len(ids)}  workgroups per CU: min {min((cu == i).sum() for i in ids)} max {max((cu == i).sum() for i in ids)}")
print("per-CU span (first start -> last stamp-10): p10 %.0f p50 %.0f p90 %.0f ticks" %
      tuple(np.percentile([end[cu == i].max() for i in ids], [10, 50, 90])))
conc, gaps, first, last = [], [], [], []
for i in ids:
    m = cu == i
    st, en = np.sort(start[m]), np.sort(end[m])
    conc.append((end[m] - start[m]).sum() / en[-1])
    first.append(st[0]); last.append(en[-1])
    # k-th end is followed by the (slots + k)-th start when the CU refills a freed slot
    slots = int((st < en[0]).sum())
    for k in range(len(st) - slots):
        gaps.append(st[slots + k] - en[k])
print("mean resident workgroups per CU over the kernel span: %.2f" % np.mean(conc))
if gaps:
    print("slot refill gap (next start - freed end, stamp 10 is before the final stores drain): mean %.0f  p50 %.0f  p90 %.0f" %
          (np.mean(gaps), np.percentile(gaps, 50), np.percentile(gaps, 90)))
