"""quick device timing of the fused tick kernel (development aid; bench.py is the contract)."""
import sys, json, time
sys.path.insert(0, ".")
import waveform_amd as wf
from tools import synth

def run(n, streams, ticks=20, hop=800, stereo=1, **kw):
    cfg = wf.Config.defaults(fft_size=n, stereo=stereo, slope=1.0, **kw)
    ring = n + hop * (ticks + 4)
    with wf.SpectrumBatch(cfg, streams, ring_frames=ring) as b:
        b.push_synth(synth.DEFAULT_SEED, 0, hop * (ticks + 2))
        b.sync()
        b.time_ticks(2, hop, hop * (ticks + 1))  # warm-up
        ms = b.time_ticks(ticks, hop, hop * (ticks - 1))
        nspec = streams * b.capture_channels
        byt = b.algorithmic_bytes_per_tick()
        print(json.dumps(dict(kernel=b.kernel_name(), streams=streams, ms=round(ms, 4), spectra_per_s=round(nspec / ms * 1e3),
                              GBps=round(byt / ms / 1e6, 1), frac_8TBps=round(byt / ms / 1e6 / 8000, 4))))

if __name__ == "__main__":
    for n, s in ((1024, 16384), (2048, 8192), (4096, 4096), (8192, 2048), (16384, 1024)):
        run(n, s)
    run(4096, 4096, stereo=0)
