"""device timing of the waveform-display tick (development aid).  usage: python tools/wave_bench.py [streams[:width] ...]"""
import sys, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import waveform_amd as wf
from waveform_amd.binding import TickParams
from tools import synth


def run(streams, width=800, ticks=30, hop=800):
    cfg = wf.Config.defaults(waveform=1, stereo=1, width=width)
    with wf.SpectrumBatch(cfg, streams, ring_frames=16384 + hop * (ticks + 4)) as b:
        b.push_synth(synth.DEFAULT_SEED, 0, hop * (ticks + 2))
        b.sync()
        # tick i looks at the audio up to hop*(i+1) frames: delay counts back from the newest sample, audio_ts is its time
        end_ns = (hop * (ticks + 2) + width) * 1_000_000_000 // 48000
        from tools.quick_bench import warm_clocks
        warm_clocks()  # the device's clocks settle after 15-20 ms of load (profiles/r02j_warmup.txt)
        ms = C.c_float(0.0)
        p = TickParams(1 / 60, hop * (ticks + 1), 0.0, 0, end_ns)
        b._ck(b.L.wf_hip_time_ticks(b.h, C.byref(p), 2, hop, C.byref(ms)))  # warm-up
        p = TickParams(1 / 60, hop * (ticks - 1), 0.0, 0, end_ns)
        b._ck(b.L.wf_hip_time_ticks(b.h, C.byref(p), ticks, hop, C.byref(ms)))
        byt = b.algorithmic_bytes_per_tick()
        print(json.dumps(dict(kernel=b.kernel_name(), streams=streams, width=width, ms=round(ms.value, 4),
                              Mrows_s=round(streams * 2 / ms.value / 1e3, 2), GBps=round(byt / ms.value / 1e6, 1),
                              frac=round(byt / ms.value / 1e6 / 8000, 4))), flush=True)


if __name__ == "__main__":
    jobs = [a.split(":") for a in sys.argv[1:]] or [("4096",), ("65536",), ("16384", "2048")]
    for j in jobs:
        run(int(j[0]), int(j[1]) if len(j) > 1 else 800)
