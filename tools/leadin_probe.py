"""How a long untimed run in front changes the wall clock of the short timed runs behind it (development aid: the driver's
20-step region is 1 ms; profiles/r06y_lead_in_ab.txt).  usage: python tools/leadin_probe.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import waveform_amd as wf
from tools import synth
import torch

HOP, N, STREAMS, DEPTH = 800, 4096, 4096, 64
cfg = wf.Config.defaults(fft_size=N, stereo=1, slope=1.0)
torch.cuda.init()
with wf.SpectrumBatch(cfg, STREAMS, ring_frames=N + HOP * (DEPTH + 4)) as b:
    b.push_synth(synth.DEFAULT_SEED, 0, HOP * (DEPTH + 2))
    b.sync()
    byt = b.algorithmic_bytes_per_tick()
    run = lambda k: b.time_ticks(k, HOP, HOP * (DEPTH - 1))
    for lead in [int(a) for a in sys.argv[1:]] or [0, 800, 2000, 4000, 800]:
        time.sleep(0.2)
        if lead:
            run(lead)
        out = []
        for i in range(8):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ms = run(20)
            torch.cuda.synchronize()
            w = (time.perf_counter() - t0) * 1e3 / 20
            out.append((round(byt / w / 1e6 / 8000, 3), round(byt / ms / 1e6 / 8000, 3)))
        print(json.dumps({"lead_in_ticks": lead, "wall/events frac of 8 consecutive 20-step runs": out}), flush=True)
