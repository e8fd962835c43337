"""development aid: per-phase s_memtime stamps of the tick kernel (needs the -DWF_PHASE_TIMING build)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, ".")
os.environ.setdefault("WF_HIP_LIB", os.path.abspath("build/variants/lib_timing.so"))
import waveform_amd as wf
from tools import synth
n, streams, hop, ticks = 4096, 4096, 800, 6
cfg = wf.Config.defaults(fft_size=n, stereo=1, slope=1.0)
b = wf.SpectrumBatch(cfg, streams, ring_frames=n + hop * (ticks + 2))
b.push_synth(synth.DEFAULT_SEED, 0, hop * (ticks + 1))
for i in range(ticks):
    b.tick(delay_frames=hop * (ticks - 1 - i))
b.sync()
nblk = streams * 2 // 2
buf = np.zeros(nblk * 16, np.uint64)
L = wf.lib()
L.wf_hip_debug_phase_clock.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
assert L.wf_hip_debug_phase_clock(b.h, buf.ctypes.data, buf.size) == 0
s = buf.reshape(nblk, 16).astype(np.int64)
d = np.diff(s[:, :11], axis=1)
names = ["fetch+nz", "facts xchg", "p1 win+pass1", "sync+p2 read", "p2 dft+write", "sync+p3 read", "p3 dft+write", "sync", "p4 split+smooth", "dB+store"]
print("stamps are s_memtime ticks (100 MHz constant clock?) -- relative shares matter")
tot = (s[:, 10] - s[:, 0])
print("block lifetime: mean %.0f  p10 %.0f  p90 %.0f" % (tot.mean(), np.percentile(tot, 10), np.percentile(tot, 90)))
for i, nm in enumerate(names):
    print(f"{nm:18s} mean {d[:, i].mean():9.1f}  ({100 * d[:, i].mean() / tot.mean():5.1f} %)   p90 {np.percentile(d[:, i], 90):9.1f}")
print("kernel span (max end - min start):", s[:, 10].max() - s[:, 0].min())
