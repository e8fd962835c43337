"""Host time of one wf_hip_tick (enqueue only) against the device time of the tick (development aid).
usage: python tools/host_cost_probe.py [N:streams ...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import waveform_amd as wf
from tools import synth

HOP, DEPTH = 800, 64
for spec in sys.argv[1:] or ["4096:4096", "2048:256", "4096:64", "16384:1024"]:
    n, streams = (int(x) for x in spec.split(":"))
    cfg = wf.Config.defaults(fft_size=n, stereo=1, slope=1.0)
    with wf.SpectrumBatch(cfg, streams, ring_frames=n + HOP * (DEPTH + 4)) as b:
        b.push_synth(synth.DEFAULT_SEED, 0, HOP * (DEPTH + 2))
        b.sync()
        b.time_ticks(400, HOP, HOP * (DEPTH - 1))
        res = []
        for k in (20, 200):
            for _ in range(3):
                b.sync()
                t0 = time.perf_counter()
                for i in range(k):
                    b.tick(delay_frames=HOP * (DEPTH - 1 - i % DEPTH))
                t1 = time.perf_counter()
                b.sync()
                t2 = time.perf_counter()
                res.append({"ticks": k, "host_us_per_tick_enqueue": round((t1 - t0) * 1e6 / k, 2), "wall_us_per_tick": round((t2 - t0) * 1e6 / k, 2)})
        dev = b.time_ticks(200, HOP, HOP * (DEPTH - 1)) * 1e3
        print(json.dumps({"shape": spec, "lanes": b.launches_per_tick(), "device_us_per_tick (events, 200 ticks in one C call)": round(dev, 2), "python loop": res}), flush=True)
