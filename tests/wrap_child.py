"""TEST HARNESS (child of tests/test_gpu_fullsize.py::test_sample_counters_wrap_at_2_to_the_32): runs against the DEVELOPMENT build of
the library (libwaveform_hip_dev.so via WF_HIP_LIB -- the release library does not export wf_hip_debug_age).
usage: python tests/wrap_child.py spectrum_normalize|meter"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import waveform_amd as wf  # noqa: E402
from tools import synth  # noqa: E402

SEED = synth.DEFAULT_SEED


def run(kind):
    if kind == "meter":
        cfg = wf.Config.defaults(meter=1, meter_rms=1, meter_ms=150)
    else:
        cfg = wf.Config.defaults(fft_size=2048, stereo=1, slope=1.0, normalize_volume=1, bars=1, interp_mode=wf.INTERP["lanczos"])
    streams, ring = 3, 1 << 17
    push = 60000 if kind != "meter" else 6000
    pushes = 50 if kind != "meter" else 40
    rings_left = 8 if kind != "meter" else 1
    age = (1 << 32) - rings_left * ring  # the wrap falls inside the run
    L = wf.lib()
    assert hasattr(L, "wf_hip_debug_age"), "needs the development build (WF_HIP_LIB=waveform_amd/libwaveform_hip_dev.so)"
    with wf.SpectrumBatch(cfg, streams, ring_frames=ring) as fresh, wf.SpectrumBatch(cfg, streams, ring_frames=ring) as old:
        if kind != "meter":
            fresh.enable_input_rms()
            old.enable_input_rms()
        assert L.wf_hip_debug_age(old.h, 0, streams, age) == 0, L.wf_hip_last_error(old.h)
        total = 0
        for i in range(pushes):
            a = synth.block(SEED, 0, streams, fresh.capture_channels, i * push, push) * np.float32(0.05 if i % 7 else 0.8)
            for b in (fresh, old):
                b.push_audio(a)
                b.tick()
            total += push
            if kind == "meter":
                assert np.array_equal(fresh.meter(), old.meter()), f"push {i}: levels differ"
            else:
                assert np.array_equal(fresh.decibels(), old.decibels()), f"push {i}: rows differ ({total} frames in, wrap at {rings_left * ring})"
                assert np.array_equal(fresh.input_rms(), old.input_rms()), f"push {i}: m_input_rms differs"
            assert np.array_equal(fresh.bars(), old.bars()) and np.array_equal(fresh.last_silent(), old.last_silent())
        assert total > rings_left * ring + 4 * push


if __name__ == "__main__":
    run(sys.argv[1])
    print("wrapped ok")
