"""Interleaved A/B of the per-tick exchange of BASELINE configs[4] at world 1 (development aid): the cfg5 per-GPU shape (8192
stereo streams, FFT 4096, 26 Lanczos bars, bars-only ticks) without any gather, with waveform_amd.dist.BarsGather writing the
send buffers from the tick kernel (wf_hip_set_bars_mirrors) and with the copy behind the tick (WF_BARS_GATHER_COPY=1); then the
C ABI's multi-device group both ways (WF_HIP_MULTI_MIRROR).  Every measurement in a process of its own, round-robin.
usage: python tools/ab_gather.py [reps]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, os, sys, time
sys.path.insert(0, %r)
import torch
import waveform_amd as wf
from waveform_amd.dist import BarsGather, shard_streams
from tools import synth
HOP, FFT, depth, streams = 800, 4096, 16, 8192
mode = sys.argv[1]
cfg = wf.Config.defaults(fft_size=FFT, stereo=1, slope=1.0, bars=1, interp_mode=wf.INTERP["lanczos"])
flags = wf.TICK_NO_DECIBELS
with wf.SpectrumBatch(cfg, streams, ring_frames=FFT + HOP * (depth + 1)) as b:
    b.push_synth(synth.DEFAULT_SEED, 0, HOP * depth)
    b.sync()
    g = BarsGather(b, shard_streams(streams, 0, 1)) if mode != "none" else None
    def run(n):
        b.time_begin()
        for i in range(n):
            b.tick(delay_frames=HOP * (depth - 1 - i %% depth), flags=flags)
            if g is not None:
                g.launch()
        ms = b.time_end()
        if g is not None:
            g.wait()
        return ms / n
    probe = run(8)
    run(int(40.0 / probe) + 1)
    best = min(run(300) for _ in range(3))
    byt = b.algorithmic_bytes_per_tick(flags)
    print(json.dumps([round(byt / best / 1e6 / 8000, 4), bool(g.zero_copy) if g is not None else None]))
''' % ROOT

if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    modes = [("no gather", "none", {}), ("BarsGather, kernel writes the send buffer", "mirror", {}), ("BarsGather, copy behind the tick", "copy", {"WF_BARS_GATHER_COPY": "1"})]
    res = {m[0]: [] for m in modes}
    for _ in range(reps):
        for name, mode, env in modes:
            r = subprocess.run([sys.executable, "-c", CHILD, mode], capture_output=True, text=True, env=dict(os.environ, **env))
            res[name].append(json.loads(r.stdout.strip().splitlines()[-1])[0] if r.returncode == 0 else "error: " + r.stderr.strip()[-200:])
    print(json.dumps({"shape": "cfg5 per-GPU shape, world 1, torch process", "frac_of_8TBps": res}), flush=True)
    res = {"wf_hip_multi, kernel writes the gathered buffer": [], "wf_hip_multi, copy behind the tick": []}
    for _ in range(reps):
        for name, env in (("wf_hip_multi, kernel writes the gathered buffer", {"WF_HIP_MULTI_MIRROR": "1"}), ("wf_hip_multi, copy behind the tick", {"WF_HIP_MULTI_MIRROR": "0"})):
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "multi_bench.py")], capture_output=True, text=True, env=dict(os.environ, **env))
            lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
            if r.returncode == 0 and lines:
                o = json.loads(lines[-1])
                res[name].append([round(o["roofline"]["frac"], 4) if "roofline" in o else None, o.get("verified")])
            else:
                res[name].append("error: " + r.stderr.strip()[-200:])
    print(json.dumps({"shape": "cfg5 per-GPU shape through wf_hip_multi_* (one device)", "frac_of_8TBps_and_verified": res}), flush=True)
