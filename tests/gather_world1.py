"""child process of tests/test_gpu_fullsize.py::test_bars_gather_world1_and_self_launching_bench (needs a GPU)"""
import sys
from pathlib import Path

import numpy as np
import torch  # before libwaveform_hip.so: one HIP runtime per process

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import waveform_amd as wf  # noqa: E402
from tools import synth  # noqa: E402
from waveform_amd.dist import BarsGather, shard_streams  # noqa: E402

torch.cuda.set_device(0)
cfg = wf.Config.defaults(fft_size=1024, stereo=1, bars=1, interp_mode=wf.INTERP["lanczos"])
streams, hop, ticks = 300, 800, 6
with wf.SpectrumBatch(cfg, streams, ring_frames=1024 + hop * (ticks + 1)) as b:
    b.push_synth(synth.DEFAULT_SEED, 0, hop * ticks)
    g = BarsGather(b, shard_streams(streams, 0, 1))
    for t in range(ticks):
        b.tick(delay_frames=hop * (ticks - 1 - t))
        k = g.launch()
        want = b.bars()  # blocking read: the reference point
        torch.cuda.synchronize()
        assert np.array_equal(g.result[k].cpu().numpy(), want), f"gather of tick {t} is not that tick's bars"
    # without host waits in between: the last gather still belongs to the last tick
    for t in range(ticks):
        b.tick(delay_frames=0)
        g.launch()
    assert np.array_equal(g.wait().cpu().numpy(), b.bars())
    assert g.zero_copy, "fft_size 1024 with bars: the tick kernel writes the send buffers itself"
    # streams the tick leaves as they are (paused), resets (hidden) and finishes, side by side: the mirror buffers -- written
    # alternately -- must carry all of them every tick (the untouched rows are copied over inside the kernel)
    mask = np.zeros(streams, np.uint8)
    mask[5:40] = 3      # WF_HIP_PAUSED
    mask[100:130] = 1   # WF_HIP_HIDDEN
    b.set_hidden(mask)
    for t in range(5):
        b.push_synth(synth.DEFAULT_SEED, hop * (ticks + t), hop)
        b.tick()
        k = g.launch()
        want = b.bars()
        torch.cuda.synchronize()
        assert np.array_equal(g.result[k].cpu().numpy(), want), f"mirror of tick {t} with paused / hidden streams"
    b.set_hidden(np.zeros(streams, np.uint8))
    for t in range(3):
        b.push_silence(hop)   # towards digital silence: rows that stop being produced
        b.tick()
        k = g.launch()
        want = b.bars()
        torch.cuda.synchronize()
        assert np.array_equal(g.result[k].cpu().numpy(), want), f"mirror of silent tick {t}"
    g.close()
# a display that comes from a kernel of its own keeps the copy behind the tick
cfg = wf.Config.defaults(fft_size=65536, stereo=1, bars=1, interp_mode=wf.INTERP["lanczos"])
with wf.SpectrumBatch(cfg, 8, ring_frames=65536 + hop * 3) as b:
    b.push_synth(synth.DEFAULT_SEED, 0, hop * 2)
    g = BarsGather(b, shard_streams(8, 0, 1))
    assert not g.zero_copy
    for t in range(2):
        b.tick(delay_frames=hop * (1 - t))
        k = g.launch()
        want = b.bars()
        torch.cuda.synchronize()
        assert np.array_equal(g.result[k].cpu().numpy(), want)
print("gather ok")
