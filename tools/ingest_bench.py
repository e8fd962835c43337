"""PCIe-inclusive rate (development aid; bench.py's `value` is measured with the audio already resident in HBM):
every step hands one 60 fps hop of host audio per stream through the C ABI (wf_hip_push_audio: H2D copy + ring append),
runs the tick and, optionally, reads the bars back.
usage: python tools/ingest_bench.py [streams] [fft]"""
import json, sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import waveform_amd as wf
from tools import synth

streams = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
hop, steps, warm = 800, 60, 10
cfg = wf.Config.defaults(fft_size=n, stereo=1, slope=1.0, bars=1, interp_mode=1)
audio = synth.block(synth.DEFAULT_SEED, 0, 1, 2, 0, hop)[0]
packet = np.ascontiguousarray(np.broadcast_to(audio[None], (streams, 2, hop)), np.float32)   # 26 MB for 4096 streams
with wf.SpectrumBatch(cfg, streams) as b:
    for mode in ("push_audio + tick", "push_audio + tick + read_bars"):
        for i in range(warm + steps):
            if i == warm:
                b.sync()
                t0 = time.perf_counter()
            b.push_audio(packet)
            b.tick()
            if mode.endswith("read_bars"):
                b.bars()
        b.sync()
        dt = (time.perf_counter() - t0) / steps
        print(json.dumps(dict(mode=mode, streams=streams, fft=n, ms_per_step=round(dt * 1e3, 3),
                              Mspectra_s=round(2 * streams / dt / 1e6, 2),
                              host_GBps=round(packet.nbytes / dt / 1e9, 2))), flush=True)

    # pipelined: page-locked buffers, the copy of packet i+1 under the tick of packet i (wf_hip_push_audio_async)
    pin = [wf.PinnedBuffer(packet.shape), wf.PinnedBuffer(packet.shape)]
    for p in pin:
        p.array[...] = packet
    out = [wf.PinnedBuffer((streams, b.display_channels, b.num_bars)), wf.PinnedBuffer((streams, b.display_channels, b.num_bars))]
    for mode in ("push_audio_async + tick", "push_audio_async + tick + read_bars", "push_audio_async + tick + read_bars_async"):
        for i in range(warm + steps):
            if i == warm:
                b.sync()
                t0 = time.perf_counter()
            slot = i & 1
            b.ingest_done(slot)            # the buffer is free again (a real host would refill it here)
            b.push_audio_async(pin[slot], streams, hop, slot)
            b.tick()
            if mode.endswith("read_bars"):
                b.bars()
            elif mode.endswith("read_bars_async"):
                b.readback_done(slot)      # the bars of two ticks ago have landed (a renderer would draw them now)
                b.read_bars_async(out[slot], slot)
        b.sync()
        dt = (time.perf_counter() - t0) / steps
        print(json.dumps(dict(mode=mode, streams=streams, fft=n, ms_per_step=round(dt * 1e3, 3),
                              Mspectra_s=round(2 * streams / dt / 1e6, 2),
                              host_GBps=round(packet.nbytes / dt / 1e9, 2))), flush=True)
