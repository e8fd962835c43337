"""quick device timing of the fused tick kernel (development aid; bench.py is the contract).
usage: python tools/quick_bench.py [N:streams ...]   (env WF_HIP_LIB selects a library build)"""
import sys, json, os
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import waveform_amd as wf
from tools import synth

def run(n, streams, ticks=30, hop=800, stereo=1, reps=3, **kw):
    cfg = wf.Config.defaults(fft_size=n, stereo=stereo, slope=1.0, **kw)
    ring = n + hop * (ticks + 4)
    with wf.SpectrumBatch(cfg, streams, ring_frames=ring) as b:
        b.push_synth(synth.DEFAULT_SEED, 0, hop * (ticks + 2))
        b.sync()
        b.time_ticks(3, hop, hop * (ticks + 1))  # warm-up
        best = min(b.time_ticks(ticks, hop, hop * (ticks - 1)) for _ in range(reps))
        nspec = streams * b.capture_channels
        byt = b.algorithmic_bytes_per_tick()
        print(json.dumps(dict(lib=os.path.basename(os.environ.get("WF_HIP_LIB", "default")), kernel=b.kernel_name(), streams=streams,
                              ms=round(best, 4), Mspectra_s=round(nspec / best / 1e3, 2), GBps=round(byt / best / 1e6, 1),
                              frac=round(byt / best / 1e6 / 8000, 4))), flush=True)

if __name__ == "__main__":
    extra = {}
    if os.environ.get("WF_BENCH_BARS"):
        extra = dict(bars=1, interp_mode=int(os.environ["WF_BENCH_BARS"]))
    if os.environ.get("WF_BENCH_CURVE"):  # curve display, interp mode as given; WF_BENCH_GAUSS=<sigma> adds the filter
        extra = dict(curve=1, interp_mode=int(os.environ["WF_BENCH_CURVE"]))
    if os.environ.get("WF_BENCH_GAUSS"):
        extra.update(filter_mode=1, filter_radius=float(os.environ["WF_BENCH_GAUSS"]))
    if os.environ.get("WF_BENCH_ROLLOFF"):  # roll-off on (q = 1, rate as given in dB per octave)
        extra.update(rolloff_q=1.0, rolloff_rate=float(os.environ["WF_BENCH_ROLLOFF"]))
    jobs = [a.split(":") for a in sys.argv[1:]] or [("1024", "16384"), ("2048", "8192"), ("4096", "4096"), ("8192", "2048"), ("16384", "1024")]
    for n, s in jobs:
        run(int(n), int(s), **extra)
