"""Interleaved A/B of the bars tail's layouts on the shapes that display bars (development aid; needs a library built with
-DWF_DEV_BUILD, e.g. tools/variant.sh dev, selected by WF_HIP_LIB): WF_HIP_BAR_PIECES=0 (bar_segments' wave-local layout:
row parked, barrier, six ds_bpermute steps) against 1 (wave-private pieces: no barrier, DPP scan, last-arriver sum).
usage: WF_HIP_LIB=variants/lib_dev.so python tools/ab_bars.py [reps]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import waveform_amd as wf
from tools import synth
from tools.quick_bench import steady_ms

HOP = 800
SHAPES = [
    ("cfg5shape 4096x8192 bars-only", dict(fft_size=4096, stereo=1, slope=1.0, bars=1, interp_mode=1), 8192, wf.TICK_NO_DECIBELS),
    ("cfg4 16384x1024 tv+bars", dict(fft_size=16384, stereo=1, tsmoothing=2, bars=1, interp_mode=1), 1024, 0),
    ("4096x4096 bars", dict(fft_size=4096, stereo=1, slope=1.0, bars=1, interp_mode=1), 4096, 0),
    ("2048x8192 bars", dict(fft_size=2048, stereo=1, slope=1.0, bars=1, interp_mode=1), 8192, 0),
    ("1024x16384 bars", dict(fft_size=1024, stereo=1, slope=1.0, bars=1, interp_mode=1), 16384, 0),
    ("8192x2048 bars", dict(fft_size=8192, stereo=1, slope=1.0, bars=1, interp_mode=1), 2048, 0),
    ("32768x512 bars", dict(fft_size=32768, stereo=1, slope=1.0, bars=1, interp_mode=1), 512, 0),
    ("plugin defaults 4096x4096 mono + Catmull-Rom curve", dict(fft_size=4096, stereo=0, slope=1.0, curve=1, interp_mode=2), 4096, 0),
    ("4096x4096 stereo + Lanczos curve", dict(fft_size=4096, stereo=1, slope=1.0, curve=1, interp_mode=1), 4096, 0),
]


def measure(kw, streams, flags, ticks=30):
    cfg = wf.Config.defaults(**kw)
    n = cfg.fft_size
    with wf.SpectrumBatch(cfg, streams, ring_frames=n + HOP * (ticks + 4)) as b:
        b.push_synth(synth.DEFAULT_SEED, 0, HOP * (ticks + 2))
        b.sync()
        ms = steady_ms(lambda k: b.time_ticks(k, HOP, HOP * (ticks - 1), flags=flags), ticks, 3)
        return ms, b.algorithmic_bytes_per_tick(flags)


if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    envs = [e.split("=") for e in os.environ.get("WF_AB_ENVS", "WF_HIP_BAR_PIECES=0,WF_HIP_BAR_PIECES=1").split(",")]
    for name, kw, streams, flags in SHAPES:
        res = {f"{k}={v}": [] for k, v in envs}
        for _ in range(reps):
            for k, v in envs:
                os.environ[k] = v
                ms, byt = measure(kw, streams, flags)
                res[f"{k}={v}"].append(round(byt / ms / 1e6 / 8000, 4))
        print(json.dumps({"shape": name, "frac_of_8TBps": res}), flush=True)
