"""TEST HARNESS: drives waveform_amd/csrc/wf_hip_multi.cpp over the mock device layer (tests/mock/mock_device.cpp) in a process
that has a sanitizer runtime preloaded (tests/test_sanitizers.py).  usage: python multi_stress.py build/libwfmulti_asan.so"""
import ctypes as C
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from waveform_amd.binding import Config, TickParams  # the structs only: the product library is not loaded

L = C.CDLL(sys.argv[1])
vp, u32, fp = C.c_void_p, C.c_uint32, C.POINTER(C.c_float)
L.wf_hip_multi_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_int), u32, u32, u32, C.POINTER(vp)]
L.wf_hip_multi_destroy.argtypes = [vp]
L.wf_hip_multi_last_error.restype = C.c_char_p
L.wf_hip_multi_last_error.argtypes = [vp]
L.wf_hip_multi_transport.restype = C.c_char_p
L.wf_hip_multi_transport.argtypes = [vp]
L.wf_hip_multi_push_synth.argtypes = [vp, u32, u32, C.c_uint64, u32, C.c_uint64, u32]
L.wf_hip_multi_tick.argtypes = [vp, C.POINTER(TickParams)]
L.wf_hip_multi_sync.argtypes = [vp]
L.wf_hip_multi_allgather_bars.argtypes = [vp]
L.wf_hip_multi_read_gathered.argtypes = [vp, u32, fp]
L.wf_hip_multi_read.argtypes = [vp, C.c_int, u32, u32, vp]
OUT_BARS, OUT_LAST_SILENT = 1, 5  # wf_hip_output
L.wf_hip_multi_set_hidden.argtypes = [vp, u32, u32, C.POINTER(C.c_uint8)]
L.wf_hip_multi_time_ticks.argtypes = [vp, C.POINTER(TickParams), u32, u32, C.c_int, fp, fp]
L.wf_hip_multi_debug_fail_next_gather.argtypes = [vp, u32]
L.wf_hip_multi_shard.restype = vp
L.wf_hip_multi_shard.argtypes = [vp, u32, C.POINTER(C.c_int), C.POINTER(u32), C.POINTER(u32)]
L.mock_bar_value.restype = C.c_float
L.mock_bar_value.argtypes = [u32, u32, u32, u32]


def expected(streams, ticks):
    s, c, b = np.meshgrid(np.arange(streams), np.arange(2), np.arange(26), indexing="ij")
    return ((s % 4096).astype(np.float32) + np.float32(0.25) * c.astype(np.float32) + b.astype(np.float32) * np.float32(0.001)
            + np.float32(5000.0) * np.float32(ticks % 64)).astype(np.float32)


def group(devices, streams):
    cfg = Config()
    cfg.fft_size, cfg.sample_rate, cfg.capture_channels, cfg.stereo, cfg.bars = 2048, 48000, 2, 1, 1
    m = vp()
    devs = (C.c_int * len(devices))(*devices)
    rc = L.wf_hip_multi_create(C.byref(cfg), devs, len(devices), streams, 0, C.byref(m))
    assert rc == 0, L.wf_hip_multi_last_error(None)
    return m


def scenario(devices, streams, rounds=6):
    m = group(devices, streams)
    try:
        assert L.wf_hip_multi_transport(m) in (b"peer", b"local"), L.wf_hip_multi_transport(m)
        assert L.wf_hip_multi_push_synth(m, 0, streams, 1, 0, 0, 800) == 0
        p = TickParams(1 / 60, 0, 0.0, 0, 0)
        out = np.empty((streams, 2, 26), np.float32)
        ticks = 0
        for _ in range(rounds):
            assert L.wf_hip_multi_tick(m, C.byref(p)) == 0
            ticks += 1
            assert L.wf_hip_multi_allgather_bars(m) == 0, L.wf_hip_multi_last_error(m)
            for i in range(len(devices)):
                assert L.wf_hip_multi_read_gathered(m, i, out.ctypes.data_as(fp)) == 0
                assert np.array_equal(out, expected(streams, ticks)), (devices, streams, i, ticks)
        ms, per = C.c_float(0), (C.c_float * len(devices))()
        assert L.wf_hip_multi_time_ticks(m, C.byref(p), 40, 0, 1, C.byref(ms), per) == 0, L.wf_hip_multi_last_error(m)
        ticks += 40
        for i in range(len(devices)):
            assert L.wf_hip_multi_read_gathered(m, i, out.ctypes.data_as(fp)) == 0
            assert np.array_equal(out, expected(streams, ticks)), "after the timed loop"
        mask = np.zeros(streams, np.uint8)
        mask[1::3] = 1
        assert L.wf_hip_multi_set_hidden(m, 0, streams, mask.ctypes.data_as(C.POINTER(C.c_uint8))) == 0
        back = np.empty(streams, np.uint8)
        assert L.wf_hip_multi_read(m, OUT_LAST_SILENT, 0, streams, back.ctypes.data_as(vp)) == 0 and np.array_equal(back, mask)
        if len(devices) > 1:
            # one shard fails inside a gather: reported, later gathers refused, everything else goes on, destroy returns
            assert L.wf_hip_multi_debug_fail_next_gather(m, len(devices) - 1) == 0
            assert L.wf_hip_multi_tick(m, C.byref(p)) == 0
            assert L.wf_hip_multi_allgather_bars(m) != 0 and b"injected" in L.wf_hip_multi_last_error(m)
            assert L.wf_hip_multi_allgather_bars(m) != 0 and b"out of service" in L.wf_hip_multi_last_error(m)
            assert L.wf_hip_multi_time_ticks(m, C.byref(p), 5, 0, 1, C.byref(ms), per) != 0
            assert L.wf_hip_multi_tick(m, C.byref(p)) == 0 and L.wf_hip_multi_sync(m) == 0
            assert L.wf_hip_multi_read(m, OUT_BARS, 0, streams, out.ctypes.data_as(vp)) == 0
        m2 = group(devices, streams)   # a failure inside the timed loop (the workers' barrier path)
        try:
            if len(devices) > 1:
                assert L.wf_hip_multi_debug_fail_next_gather(m2, 0) == 0
                assert L.wf_hip_multi_time_ticks(m2, C.byref(p), 8, 0, 1, C.byref(ms), per) != 0 and b"injected" in L.wf_hip_multi_last_error(m2)
                assert L.wf_hip_multi_sync(m2) == 0
        finally:
            L.wf_hip_multi_destroy(m2)
    finally:
        L.wf_hip_multi_destroy(m)


if __name__ == "__main__":
    os.environ["WF_HIP_MULTI_TRANSPORT"] = "peer"
    for devices, streams in (([0], 9), ([0, 1, 2, 3], 64), ([0, 1, 2, 3], 67), ([0, 0, 1], 10), (list(range(4)) * 2, 8 * 33 + 5)):
        scenario(devices, streams)
    # distinct groups on distinct host threads at the same time
    errs = []

    def run(devs, n):
        try:
            scenario(devs, n, rounds=12)
        except BaseException as e:  # noqa
            errs.append(repr(e))

    th = [threading.Thread(target=run, args=(d, n)) for d, n in (([0, 1], 32), ([2, 3], 45), ([0, 1, 2, 3], 16))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    print("multi stress ok")
