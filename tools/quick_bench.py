"""quick device timing of the fused tick kernel, steady state (development aid; bench.py is the contract).
usage: python tools/quick_bench.py [N:streams ...]   (env WF_HIP_LIB selects a library build)"""
import sys, json, os
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import waveform_amd as wf
from tools import synth

WARM_MS, TIMED_MS = 40.0, 25.0


def steady_ms(time_fn, min_ticks=30, reps=3):
    """time_fn(n) -> average device ms per tick over n back-to-back ticks.  The device's clocks settle after 15-20 ms of load
    (profiles/r02j_warmup.txt): a lead-in of WARM_MS, then the best of `reps` regions of at least TIMED_MS / min_ticks ticks."""
    probe = time_fn(8)
    time_fn(int(WARM_MS / probe) + 1)
    n = max(min_ticks, int(TIMED_MS / probe) + 1)
    return min(time_fn(n) for _ in range(reps))


def warm_clocks(ms=WARM_MS, hop=800):
    """for timings that cannot walk their resident audio twice (meter, waveform: their ticks consume it): WARM_MS of spectrum
    ticks on a batch of its own right before"""
    cfg = wf.Config.defaults(fft_size=4096, stereo=1)
    with wf.SpectrumBatch(cfg, 2048, ring_frames=4096 + hop * 20) as b:
        b.push_synth(synth.DEFAULT_SEED, 0, hop * 16)
        probe = b.time_ticks(8, hop, hop * 15)
        b.time_ticks(int(ms / probe) + 1, hop, hop * 15)


def run(n, streams, ticks=30, hop=800, stereo=1, reps=3, **kw):
    cfg = wf.Config.defaults(fft_size=n, stereo=stereo, slope=1.0, **kw)
    ring = n + hop * (ticks + 4)
    with wf.SpectrumBatch(cfg, streams, ring_frames=ring) as b:
        b.push_synth(synth.DEFAULT_SEED, 0, hop * (ticks + 2))
        b.sync()
        best = steady_ms(lambda k: b.time_ticks(k, hop, hop * (ticks - 1)), ticks, reps)  # (the walk over the resident audio wraps)
        nspec = streams * b.capture_channels
        byt = b.algorithmic_bytes_per_tick()
        print(json.dumps(dict(lib=os.path.basename(os.environ.get("WF_HIP_LIB", "default")), kernel=b.kernel_name(), streams=streams,
                              ms=round(best, 4), algorithmic_bytes_per_tick=int(byt), Mspectra_s=round(nspec / best / 1e3, 2), GBps=round(byt / best / 1e6, 1),
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
