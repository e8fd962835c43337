"""device timing of the level-meter tick (development aid).  usage: python tools/meter_bench.py [streams[:meter_ms[:rms]] ...]"""
import sys, json
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import waveform_amd as wf
from tools import synth


def run(streams, meter_ms=150, rms=1, ticks=30, hop=800, reps=3, **kw):
    cfg = wf.Config.defaults(meter=1, meter_ms=meter_ms, meter_rms=rms, **kw)
    size = (48000 * meter_ms // 1000) & -16
    ring = size + hop * (ticks + 4)
    with wf.SpectrumBatch(cfg, streams, ring_frames=ring) as b:
        assert b.fft_size == size
        b.push_synth(synth.DEFAULT_SEED, 0, hop * (ticks + 2))
        b.sync()
        from tools.quick_bench import warm_clocks
        warm_clocks()  # the device's clocks settle after 15-20 ms of load (profiles/r02j_warmup.txt)
        b.time_ticks(3, hop, hop * (ticks + 1))  # warm-up
        best = min(b.time_ticks(ticks, hop, hop * (ticks - 1)) for _ in range(reps))
        byt = b.algorithmic_bytes_per_tick()
        print(json.dumps(dict(kernel=b.kernel_name(), streams=streams, size=size, rms=rms, ms=round(best, 4),
                              Mlevels_s=round(streams * b.capture_channels / best / 1e3, 2), GBps=round(byt / best / 1e6, 1),
                              frac=round(byt / best / 1e6 / 8000, 4))), flush=True)


if __name__ == "__main__":
    jobs = [a.split(":") for a in sys.argv[1:]] or [("4096",), ("16384",), ("65536",), ("16384", "150", "0"), ("16384", "500")]
    for j in jobs:
        run(int(j[0]), int(j[1]) if len(j) > 1 else 150, int(j[2]) if len(j) > 2 else 1)
