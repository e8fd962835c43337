import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["WF_HIP_MULTI_TRANSPORT"] = "peer"
import waveform_amd as wf
HOP, FFT, SEED = 800, 4096, 0x5741564546524D31
cfg = wf.Config.defaults(fft_size=FFT, stereo=1, slope=1.0, tsmoothing=wf.TSMOOTH["exponential"], gravity=0.65, bars=1, interp_mode=wf.INTERP["lanczos"])
flags, depth = wf.TICK_NO_DECIBELS, 16
with wf.MultiBatch(cfg, 8192, [0, 0], ring_frames=FFT + HOP * (depth + 1)) as m:
    m.push_synth(SEED, 0, HOP * depth)
    m.sync()
    first = HOP * (depth - 1)
    m.time_ticks(400, HOP, first, gather=True, flags=flags)
    ms, _ = m.time_ticks(60, HOP, first, gather=True, flags=flags)
    print("ms per tick with gather", ms)
