"""experiment (development aid): one batch of 4096 streams vs two concurrent half-batches (two handles = two HIP streams)"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import waveform_amd as wf
from waveform_amd.binding import TickParams
from tools import synth

def make(streams, ticks, hop=800, n=4096):
    cfg = wf.Config.defaults(fft_size=n, stereo=1, slope=1.0)
    b = wf.SpectrumBatch(cfg, streams, ring_frames=n + hop * (ticks + 4))
    b.push_synth(synth.DEFAULT_SEED, 0, hop * (ticks + 2))
    b.sync()
    return b

def run(batches, ticks, hop=800):
    for b in batches:
        b.tick(delay_frames=hop * (ticks + 1)); b.sync()
    t0 = time.perf_counter()
    for i in range(ticks):
        for b in batches:
            b.tick(delay_frames=hop * (ticks - 1 - i))
    for b in batches:
        b.sync()
    return (time.perf_counter() - t0) / ticks * 1e6

ticks = 200
one = [make(4096, ticks)]
print("one handle x 4096 streams: %.1f us per tick" % min(run(one, ticks) for _ in range(3)))
one[0].close()
two = [make(2048, ticks), make(2048, ticks)]
print("two handles x 2048 streams: %.1f us per tick" % min(run(two, ticks) for _ in range(3)))
four = two + [make(1024, ticks)]
