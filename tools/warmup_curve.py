"""How long the device needs before a timing is a steady-state timing: the headline shape ticked back to back from a cold
start, average device time per tick over consecutive chunks of 50 ticks (HIP events; every chunk ends with a drained device, which costs it ~2 %).
usage: python tools/warmup_curve.py [chunks]     -> one line per chunk: first tick, ms per tick, fraction of the HBM peak"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import waveform_amd as wf
from tools import synth

if __name__ == "__main__":
    chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    hop, depth, n = 800, 64, 50
    cfg = wf.Config.defaults(fft_size=4096, stereo=1, slope=1.0, window=wf.WINDOW["hann"], tsmoothing=wf.TSMOOTH["exponential"], gravity=0.65)
    with wf.SpectrumBatch(cfg, 4096, ring_frames=4096 + hop * (depth + 1)) as b:
        b.push_synth(synth.DEFAULT_SEED, 0, hop * depth)
        b.sync()
        time.sleep(0.5)  # an idle device
        byt = b.algorithmic_bytes_per_tick()
        t = 0.0
        for c in range(chunks):
            ms = b.time_ticks(n, hop, hop * (depth - 1))
            print(f"ticks {c * n:5d}-{c * n + n - 1:5d}  {t:7.2f} ms in  {ms * 1e3:7.2f} us per tick  {byt / ms / 1e6 / 8000:.3f} of 8 TB/s", flush=True)
            t += ms * n
