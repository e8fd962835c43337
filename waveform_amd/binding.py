"""ctypes binding of libwaveform_hip.so (include/wf_hip.h).  Thin by design: argument
marshalling only, no numerics and no fallbacks."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = None

WINDOW = dict(none=0, hann=1, hamming=2, blackman=3, blackman_harris=4, power_of_sine=5)
TSMOOTH = dict(none=0, exponential=1, tvexponential=2)
INTERP = dict(point=0, lanczos=1, catrom=2)


class WfHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"wf_hip error {code}: {msg}")
        self.code = code


class Config(C.Structure):
    """struct wf_config (include/wf_config.h)."""
    _fields_ = [
        ("fft_size", C.c_uint32), ("sample_rate", C.c_uint32), ("capture_channels", C.c_uint32), ("stereo", C.c_uint32),
        ("window", C.c_int32), ("sine_exponent", C.c_int32), ("tsmoothing", C.c_int32), ("gravity", C.c_float),
        ("fast_peaks", C.c_uint32), ("slope", C.c_float), ("rolloff_q", C.c_float), ("rolloff_rate", C.c_float),
        ("cutoff_low", C.c_int32), ("cutoff_high", C.c_int32), ("floor_db", C.c_int32), ("ceiling_db", C.c_int32),
        ("normalize_volume", C.c_uint32), ("volume_target", C.c_float), ("max_gain", C.c_float),
        ("bars", C.c_uint32), ("interp_mode", C.c_int32), ("log_scale", C.c_uint32), ("mirror_freq_axis", C.c_uint32),
        ("width", C.c_uint32), ("height", C.c_uint32), ("bar_width", C.c_int32), ("bar_gap", C.c_int32),
        ("channel_spacing", C.c_int32), ("min_bar_height", C.c_int32), ("rounded_caps", C.c_uint32),
        ("curve", C.c_uint32), ("filter_mode", C.c_int32), ("filter_radius", C.c_float),
        ("meter", C.c_uint32), ("meter_rms", C.c_uint32), ("meter_ms", C.c_int32),
        ("waveform", C.c_uint32), ("vertices", C.c_uint32), ("step_width", C.c_int32), ("step_gap", C.c_int32),
        ("radial", C.c_uint32),
    ]

    @classmethod
    def defaults(cls, **overrides) -> "Config":
        cfg = cls()
        lib().wf_config_defaults(C.byref(cfg))
        for k, v in overrides.items():
            if not hasattr(cfg, k):
                raise AttributeError(f"wf_config has no field {k!r}")
            setattr(cfg, k, v)
        return cfg


class TickParams(C.Structure):
    _fields_ = [("seconds", C.c_float), ("delay_frames", C.c_uint32), ("input_rms", C.c_float), ("flags", C.c_uint32),
                ("audio_ts_ns", C.c_uint64)]


TICK_NO_DECIBELS = 1


# wf_hip_output / wf_hip_table_id (include/wf_hip.h)
OUT_DECIBELS, OUT_BARS, OUT_PREMIRROR, OUT_VERTICES, OUT_VERTEX_COUNTS, OUT_LAST_SILENT, OUT_TSMOOTH, OUT_METER, OUT_INPUT_RMS, OUT_WAVEFORM_TS = range(10)
(TABLE_WINDOW, TABLE_WINDOW_SUM, TABLE_SLOPE, TABLE_ROLLOFF, TABLE_INTERP_INDICES, TABLE_BAND_WIDTHS, TABLE_INTERP_WEIGHTS,
 TABLE_INTERP_SHAPE) = range(8)


class Readback(C.Structure):
    """wf_hip_readback: the page-locked destinations of one wf_hip_read_async (NULL leaves an output out)"""
    _fields_ = [("rows", C.c_void_p), ("last_silent", C.c_void_p), ("bars", C.c_void_p), ("premirror", C.c_void_p), ("vertices", C.c_void_p),
                ("vertex_counts", C.c_void_p), ("input_rms", C.c_void_p), ("meter", C.c_void_p)]


def library_path() -> Path:
    # WF_HIP_LIB: development aid for A/B-ing kernel builds; the default is the in-tree library
    import os
    override = os.environ.get("WF_HIP_LIB")
    return Path(override) if override else _HERE / "libwaveform_hip.so"


def lib():
    """Loads libwaveform_hip.so; raises if it has not been built (no fallback)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    p = library_path()
    if not p.exists():
        raise FileNotFoundError(f"{p} not built: run `make -C waveform_amd/csrc` (or __graft_entry__.build())")
    L = C.CDLL(str(p))
    vp, u32, u64 = C.c_void_p, C.c_uint32, C.c_uint64
    fp = C.POINTER(C.c_float)
    L.wf_config_defaults.argtypes = [C.POINTER(Config)]
    L.wf_hip_abi_version.restype = C.c_int
    L.wf_hip_device_count.restype = C.c_int
    L.wf_hip_last_error.restype = C.c_char_p
    L.wf_hip_last_error.argtypes = [vp]
    L.wf_hip_create.argtypes = [C.POINTER(Config), C.c_int, u32, u32, C.POINTER(vp)]
    L.wf_hip_destroy.argtypes = [vp]
    L.wf_hip_reset.argtypes = [vp, u32, u32]
    for n in ("fft_size", "num_streams", "capture_channels", "output_channels", "display_channels", "num_bars", "ring_frames"):
        f = getattr(L, "wf_hip_" + n)
        f.restype = u32
        f.argtypes = [vp]
    L.wf_hip_push_audio.argtypes = [vp, u32, u32, fp, u32]
    L.wf_hip_push_audio_device.argtypes = [vp, u32, u32, vp, u32]
    L.wf_hip_push_audio_async.argtypes = [vp, u32, u32, vp, u32, u32]
    L.wf_hip_ingest_done.argtypes = [vp, u32]
    L.wf_hip_read_async.argtypes = [vp, u32, u32, C.POINTER(Readback), u32]
    L.wf_hip_readback_done.argtypes = [vp, u32]
    L.wf_hip_host_alloc.restype = vp
    L.wf_hip_host_alloc.argtypes = [C.c_size_t]
    L.wf_hip_host_free.argtypes = [vp]
    L.wf_hip_push_synth.argtypes = [vp, u32, u32, u64, u32, u64, u32]
    L.wf_hip_push_audio_muted.argtypes = [vp, u32, u32, fp, u32]
    L.wf_hip_enable_input_rms.argtypes = [vp, C.c_int]
    L.wf_hip_tick.argtypes = [vp, C.POINTER(TickParams)]
    L.wf_hip_set_hidden.argtypes = [vp, u32, u32, C.POINTER(C.c_uint8)]
    L.wf_hip_set_input_rms.argtypes = [vp, u32, u32, fp]
    L.wf_hip_set_stream_delay.argtypes = [vp, u32, u32, C.POINTER(C.c_uint32)]
    L.wf_hip_set_stream_audio_ts.argtypes = [vp, u32, u32, C.POINTER(C.c_uint64)]
    L.wf_hip_sync.argtypes = [vp]
    L.wf_hip_read.argtypes = [vp, C.c_int, u32, u32, vp]
    L.wf_hip_output_bytes.restype = C.c_size_t
    L.wf_hip_output_bytes.argtypes = [vp, C.c_int]
    L.wf_hip_num_vertices.restype = u32
    L.wf_hip_num_vertices.argtypes = [vp]
    L.wf_hip_copy_bars_device_async.argtypes = [vp, u32, u32, vp, vp]
    L.wf_hip_wait_event.argtypes = [vp, vp]
    L.wf_hip_time_begin.argtypes = [vp]
    L.wf_hip_time_end.argtypes = [vp, fp]
    L.wf_hip_write_tsmooth.argtypes = [vp, u32, u32, fp]
    for n in ("decibels_device", "bars_device", "stream"):
        f = getattr(L, "wf_hip_" + n)
        f.restype = vp
        f.argtypes = [vp]
    L.wf_hip_table.restype = C.c_size_t
    L.wf_hip_table.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.wf_hip_gravity.restype = C.c_float
    L.wf_hip_gravity.argtypes = [vp, C.c_float]
    L.wf_hip_db_min.restype = C.c_float
    L.wf_hip_time_ticks.argtypes = [vp, C.POINTER(TickParams), u32, u32, fp]
    L.wf_hip_kernel_name.restype = C.c_char_p
    L.wf_hip_kernel_name.argtypes = [vp]
    if hasattr(L, "wf_hip_debug_age"):  # development builds only (libwaveform_hip_dev.so, -DWF_DEV_BUILD)
        L.wf_hip_debug_age.argtypes = [vp, u32, u32, u32]
    L.wf_hip_launches_per_tick.restype = u32
    L.wf_hip_launches_per_tick.argtypes = [vp]
    L.wf_hip_algorithmic_bytes_per_tick.restype = u64
    L.wf_hip_algorithmic_bytes_per_tick.argtypes = [vp, u32]
    L.wf_hip_set_bars_mirrors.argtypes = [vp, u32, C.POINTER(vp), C.POINTER(vp)]
    L.wf_hip_bars_mirror_ready.argtypes = [vp, vp, C.POINTER(vp)]
    # one batch over several devices (wf_hip_multi_*)
    L.wf_hip_multi_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_int), u32, u32, u32, C.POINTER(vp)]
    L.wf_hip_multi_destroy.argtypes = [vp]
    L.wf_hip_multi_last_error.restype = C.c_char_p
    L.wf_hip_multi_last_error.argtypes = [vp]
    L.wf_hip_multi_transport.restype = C.c_char_p
    L.wf_hip_multi_transport.argtypes = [vp]
    for n in ("num_devices", "num_streams"):
        f = getattr(L, "wf_hip_multi_" + n)
        f.restype = u32
        f.argtypes = [vp]
    L.wf_hip_multi_shard.restype = vp
    L.wf_hip_multi_shard.argtypes = [vp, u32, C.POINTER(C.c_int), C.POINTER(u32), C.POINTER(u32)]
    L.wf_hip_multi_push_audio.argtypes = [vp, u32, u32, fp, u32]
    L.wf_hip_multi_push_synth.argtypes = [vp, u32, u32, u64, u32, u64, u32]
    L.wf_hip_multi_set_hidden.argtypes = [vp, u32, u32, C.POINTER(C.c_uint8)]
    L.wf_hip_multi_reset.argtypes = [vp, u32, u32]
    L.wf_hip_multi_tick.argtypes = [vp, C.POINTER(TickParams)]
    L.wf_hip_multi_sync.argtypes = [vp]
    L.wf_hip_multi_read.argtypes = [vp, C.c_int, u32, u32, vp]
    L.wf_hip_multi_allgather_bars.argtypes = [vp]
    L.wf_hip_multi_gathered_device.restype = vp
    L.wf_hip_multi_gathered_device.argtypes = [vp, u32]
    L.wf_hip_multi_gather_stream.restype = vp
    L.wf_hip_multi_gather_stream.argtypes = [vp, u32]
    L.wf_hip_multi_read_gathered.argtypes = [vp, u32, fp]
    L.wf_hip_multi_time_ticks.argtypes = [vp, C.POINTER(TickParams), u32, u32, C.c_int, fp, fp]
    if hasattr(L, "wf_hip_multi_debug_fail_next_gather"):  # development builds only
        L.wf_hip_multi_debug_fail_next_gather.argtypes = [vp, u32]
    _LIB = L
    return L


def device_count() -> int:
    return int(lib().wf_hip_device_count())


def db_min() -> float:
    return float(lib().wf_hip_db_min())


def _copy(ptr, n, dtype=np.float32):
    if not ptr or n == 0:
        return None
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


class SpectrumBatch:
    """A batch of `streams` independent sources sharing one configuration, resident on one GPU."""

    def __init__(self, cfg: Config, streams: int, device: int = 0, ring_frames: int = 0):
        self.L = lib()
        self.cfg = cfg
        h = C.c_void_p()
        rc = self.L.wf_hip_create(C.byref(cfg), device, streams, ring_frames, C.byref(h))
        if rc != 0:
            raise WfHipError(rc, self.L.wf_hip_last_error(None).decode())
        self.h = h
        self.streams = streams
        self.fft_size = self.L.wf_hip_fft_size(h)
        self.bins = self.fft_size if cfg.waveform else self.fft_size // 2  # floats per m_decibels row
        self.capture_channels = self.L.wf_hip_capture_channels(h)
        self.output_channels = self.L.wf_hip_output_channels(h)
        self.display_channels = self.L.wf_hip_display_channels(h)
        self.num_bars = self.L.wf_hip_num_bars(h)
        self.ring_frames = self.L.wf_hip_ring_frames(h)

    # -- plumbing -----------------------------------------------------------------
    def _ck(self, rc):
        if rc != 0:
            raise WfHipError(rc, self.L.wf_hip_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.wf_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- audio ----------------------------------------------------------------------
    def push_audio(self, samples: np.ndarray, first: int = 0):
        """samples: float32 [count, capture_channels, frames]"""
        s = np.ascontiguousarray(samples, dtype=np.float32)
        assert s.ndim == 3 and s.shape[1] == self.capture_channels, s.shape
        self._ck(self.L.wf_hip_push_audio(self.h, first, s.shape[0], s.ctypes.data_as(C.POINTER(C.c_float)), s.shape[2]))

    def push_audio_muted(self, samples: np.ndarray, first: int = 0):
        """muted packet: zeros into the rings, `samples` into the device RMS producer"""
        s = np.ascontiguousarray(samples, dtype=np.float32)
        assert s.ndim == 3 and s.shape[1] == self.capture_channels, s.shape
        self._ck(self.L.wf_hip_push_audio_muted(self.h, first, s.shape[0], s.ctypes.data_as(C.POINTER(C.c_float)), s.shape[2]))

    def enable_input_rms(self):
        """update_input_rms on the device from now on (cfg.normalize_volume)"""
        self._ck(self.L.wf_hip_enable_input_rms(self.h, 0))

    def input_rms(self, first: int = 0, count: int | None = None) -> np.ndarray:
        count = self.streams - first if count is None else count
        return self._read(OUT_INPUT_RMS, first, count, (), np.float32)

    def push_audio_async(self, pinned: "PinnedBuffer", count: int, frames: int, slot: int, first: int = 0):
        """pipelined ingest from page-locked memory (see wf_hip_push_audio_async); does not wait"""
        self._ck(self.L.wf_hip_push_audio_async(self.h, first, count, C.c_void_p(pinned.ptr), frames, slot))

    def read_bars_async(self, pinned: "PinnedBuffer", slot: int, first: int = 0, count: int | None = None):
        """bars of the ticks enqueued so far -> page-locked memory, without waiting (wf_hip_read_async, the bars alone)"""
        count = self.streams - first if count is None else count
        dst = Readback(bars=pinned.ptr)
        self._ck(self.L.wf_hip_read_async(self.h, first, count, C.byref(dst), slot))

    def readback_done(self, slot: int):
        self._ck(self.L.wf_hip_readback_done(self.h, slot))

    def ingest_done(self, slot: int):
        self._ck(self.L.wf_hip_ingest_done(self.h, slot))

    def push_audio_device(self, dev_ptr: int, count: int, frames: int, first: int = 0):
        self._ck(self.L.wf_hip_push_audio_device(self.h, first, count, C.c_void_p(dev_ptr), frames))

    def push_synth(self, seed: int, index0: int, frames: int, first: int = 0, count: int | None = None, stream_id0: int = 0):
        count = self.streams - first if count is None else count
        self._ck(self.L.wf_hip_push_synth(self.h, first, count, seed, stream_id0, index0, frames))

    def push_silence(self, frames: int, first: int = 0, count: int | None = None):
        count = self.streams - first if count is None else count
        self._ck(self.L.wf_hip_push_audio_muted(self.h, first, count, None, frames))  # a packet without data: zeros

    def reset(self, first: int = 0, count: int | None = None):
        count = self.streams - first if count is None else count
        self._ck(self.L.wf_hip_reset(self.h, first, count))

    # -- tick -------------------------------------------------------------------------
    def tick(self, seconds: float = 1.0 / 60.0, delay_frames: int = 0, input_rms: float = 0.0, flags: int = 0, audio_ts_ns: int = 0):
        p = TickParams(seconds, delay_frames, input_rms, flags, audio_ts_ns)
        self._ck(self.L.wf_hip_tick(self.h, C.byref(p)))

    def set_hidden(self, mask, first: int = 0):
        """mask: uint8[count]; non-zero = hidden (1) / capture timed out (2) (reset branch of tick_spectrum; tick_meter
        tells the two apart)"""
        m = np.ascontiguousarray(mask, dtype=np.uint8)
        self._ck(self.L.wf_hip_set_hidden(self.h, first, len(m), m.ctypes.data_as(C.POINTER(C.c_uint8))))

    def set_stream_delay(self, delay_frames, first: int = 0):
        """delay_frames: uint32[count], A/V-sync delay of streams first.. in frames (added to the tick's delay_frames)"""
        d = np.ascontiguousarray(delay_frames, dtype=np.uint32)
        self._ck(self.L.wf_hip_set_stream_delay(self.h, first, len(d), d.ctypes.data_as(C.POINTER(C.c_uint32))))

    def set_stream_audio_ts(self, audio_ts_ns, first: int = 0):
        """waveform batches: m_audio_ts per stream (ns) instead of TickParams.audio_ts_ns (wf_hip_set_stream_audio_ts)"""
        d = np.ascontiguousarray(audio_ts_ns, dtype=np.uint64)
        self._ck(self.L.wf_hip_set_stream_audio_ts(self.h, first, len(d), d.ctypes.data_as(C.POINTER(C.c_uint64))))

    def set_input_rms(self, rms, first: int = 0):
        """rms: float32[count], m_input_rms of streams first.. (per-stream volume normalisation)"""
        r = np.ascontiguousarray(rms, dtype=np.float32)
        self._ck(self.L.wf_hip_set_input_rms(self.h, first, len(r), r.ctypes.data_as(C.POINTER(C.c_float))))

    def sync(self):
        self._ck(self.L.wf_hip_sync(self.h))

    def time_ticks(self, ticks: int, hop: int, first_delay: int, seconds: float = 1.0 / 60.0, flags: int = 0) -> float:
        """average fused-kernel duration in ms over `ticks` back-to-back ticks (hipEvents on the handle's stream)"""
        p = TickParams(seconds, first_delay, 0.0, flags, 0)
        ms = C.c_float(0.0)
        self._ck(self.L.wf_hip_time_ticks(self.h, C.byref(p), ticks, hop, C.byref(ms)))
        return float(ms.value)

    # -- results ------------------------------------------------------------------------
    def decibels(self, first: int = 0, count: int | None = None) -> np.ndarray:
        count = self.streams - first if count is None else count
        return self._read(OUT_DECIBELS, first, count, (self.output_channels, self.bins), np.float32)

    def _read(self, what: int, first: int, count: int | None, shape, dtype) -> np.ndarray:
        """wf_hip_read: output `what` of streams [first, first+count) as [count, *shape]"""
        count = self.streams - first if count is None else count
        out = np.empty((count,) + tuple(shape), dtype)
        per = int(self.L.wf_hip_output_bytes(self.h, what))
        assert per == 0 or per * count == out.nbytes, (what, per, out.shape)  # (0: the call below reports why the batch has none)
        self._ck(self.L.wf_hip_read(self.h, what, first, count, out.ctypes.data_as(C.c_void_p)))
        return out

    def bars(self, first: int = 0, count: int | None = None) -> np.ndarray:
        return self._read(OUT_BARS, first, count, (self.display_channels, self.num_bars), np.float32)

    def premirror(self, first: int = 0, count: int | None = None) -> np.ndarray:
        """mirrored displays: the one value the outputs above the middle had before the mirror, [count, display_channels]"""
        return self._read(OUT_PREMIRROR, first, count, (self.display_channels,), np.float32)

    def copy_bars_to_device_async(self, dev_ptr: int, consumer_stream: int, first: int = 0, count: int | None = None):
        """the same without waiting: `consumer_stream` (a hipStream_t handle, e.g. torch.cuda.Stream.cuda_stream) is made
        to wait for the copy; the handle goes on with its next tick"""
        count = self.streams - first if count is None else count
        self._ck(self.L.wf_hip_copy_bars_device_async(self.h, first, count, C.c_void_p(dev_ptr), C.c_void_p(consumer_stream)))

    def set_bars_mirrors(self, set0, set1):
        """from the next tick on every tick also leaves the whole batch's bars in every buffer of the current write set (set0 /
        set1: sequences of up to 8 device pointers each -- buffers of [streams][display_channels][num_bars] floats owned by the
        caller); bars_mirror_ready() hands the write set over and switches to the other.  Empty sequences turn it off (the call
        waits for the ticks in flight: the old buffers may be freed afterwards).  Raises WfHipError (code -2,
        WF_HIP_ERR_UNSUPPORTED) for fft sizes that are not powers of two and for batches whose display comes from a kernel of
        its own."""
        n = len(set0)
        assert n == len(set1) and n <= 8
        a0 = (C.c_void_p * max(n, 1))(*set0)
        a1 = (C.c_void_p * max(n, 1))(*set1)
        self._ck(self.L.wf_hip_set_bars_mirrors(self.h, n, a0, a1))

    def bars_mirror_ready(self, consumer_stream: int) -> int:
        """hand-over: `consumer_stream` waits for the newest tick; returns the device pointer of buffer 0 of the set the ticks
        have written (filled from the handle's own bars if no tick has); the other set becomes the ticks' target"""
        out = C.c_void_p(0)
        self._ck(self.L.wf_hip_bars_mirror_ready(self.h, C.c_void_p(consumer_stream), C.byref(out)))
        return out.value

    def time_begin(self):
        self._ck(self.L.wf_hip_time_begin(self.h))

    def time_end(self) -> float:
        """device milliseconds since time_begin (everything the handle issued in between, on every lane)"""
        ms = C.c_float(0.0)
        self._ck(self.L.wf_hip_time_end(self.h, C.byref(ms)))
        return float(ms.value)

    def meter(self, first: int = 0, count: int | None = None) -> np.ndarray:
        """meter batches: m_meter_val in dBFS, [count, capture_channels]"""
        count = self.streams - first if count is None else count
        return self._read(OUT_METER, first, count, (self.capture_channels,), np.float32)

    def tsmooth(self, first: int = 0, count: int | None = None) -> np.ndarray:
        return self._read(OUT_TSMOOTH, first, count, (self.capture_channels, self.bins), np.float32)

    def set_tsmooth(self, state: np.ndarray, first: int = 0):
        s = np.ascontiguousarray(state, dtype=np.float32)
        self._ck(self.L.wf_hip_write_tsmooth(self.h, first, s.shape[0], s.ctypes.data_as(C.POINTER(C.c_float))))

    def last_silent(self, first: int = 0, count: int | None = None) -> np.ndarray:
        return self._read(OUT_LAST_SILENT, first, count, (), np.uint8).astype(bool)

    def waveform_ts(self, first: int = 0, count: int | None = None) -> np.ndarray:
        """m_waveform_ts per stream (ns) as the last enqueued tick leaves it (waveform batches)"""
        return self._read(OUT_WAVEFORM_TS, first, count, (), np.uint64)

    def decibels_device_ptr(self) -> int:
        return int(self.L.wf_hip_decibels_device(self.h) or 0)

    def vertices(self, first: int = 0, count: int | None = None) -> np.ndarray:
        """[count, display_channels, num_vertices, 4]: what render_bars / render_curve hand to gs_draw (cfg.vertices)"""
        n = int(self.L.wf_hip_num_vertices(self.h))
        return self._read(OUT_VERTICES, first, count, (self.display_channels, n, 4), np.float32)

    def vertex_counts(self, first: int = 0, count: int | None = None) -> np.ndarray:
        """[count, display_channels]: vertices each row's draw call uses (constant unless the bars are stepped)"""
        return self._read(OUT_VERTEX_COUNTS, first, count, (self.display_channels,), np.uint32)

    def bars_device_ptr(self) -> int:
        return int(self.L.wf_hip_bars_device(self.h) or 0)

    def stream_ptr(self) -> int:
        return int(self.L.wf_hip_stream(self.h) or 0)

    # -- tables / measurement ---------------------------------------------------------------
    def _table(self, which: int, ctype, dtype):
        p = C.c_void_p()
        n = int(self.L.wf_hip_table(self.h, which, C.byref(p)))
        return _copy(C.cast(p, C.POINTER(ctype)), n, dtype) if p.value else None

    def table_window(self):
        return self._table(TABLE_WINDOW, C.c_float, np.float32), float(self._table(TABLE_WINDOW_SUM, C.c_float, np.float32)[0])

    def table(self, name: str):
        if name == "band_widths":
            return self._table(TABLE_BAND_WIDTHS, C.c_int, np.int32)
        if name == "interp_weights":
            r, t = self._table(TABLE_INTERP_SHAPE, C.c_int, np.int32)
            return self._table(TABLE_INTERP_WEIGHTS, C.c_float, np.float32), int(r), int(t)
        which = {"slope": TABLE_SLOPE, "rolloff": TABLE_ROLLOFF, "interp_indices": TABLE_INTERP_INDICES}[name]
        return self._table(which, C.c_float, np.float32)

    def gravity(self, seconds: float) -> float:
        return float(self.L.wf_hip_gravity(self.h, seconds))

    def kernel_name(self) -> str:
        return self.L.wf_hip_kernel_name(self.h).decode()

    def launches_per_tick(self) -> int:
        return int(self.L.wf_hip_launches_per_tick(self.h))

    def algorithmic_bytes_per_tick(self, flags: int = 0) -> int:
        return int(self.L.wf_hip_algorithmic_bytes_per_tick(self.h, flags))


class PinnedBuffer:
    """page-locked host memory (wf_hip_host_alloc) viewed as a float32 numpy array"""

    def __init__(self, shape):
        self.L = lib()
        n = int(np.prod(shape))
        self.ptr = self.L.wf_hip_host_alloc(n * 4)
        if not self.ptr:
            raise MemoryError("wf_hip_host_alloc failed")
        self.array = np.ctypeslib.as_array(C.cast(self.ptr, C.POINTER(C.c_float)), shape=(n,)).reshape(shape)

    def close(self):
        if getattr(self, "ptr", None):
            self.array = None
            self.L.wf_hip_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiBatch:
    """One batch of `streams` sources sharded contiguously over several devices of one node, single process, one host thread
    per device (wf_hip_multi_*); all stream indices are global.  allgather_bars() leaves every stream's bars on every device."""

    def __init__(self, cfg: Config, streams: int, devices, ring_frames: int = 0):
        self.L = lib()
        self.cfg = cfg
        devs = (C.c_int * len(devices))(*devices)
        m = C.c_void_p()
        rc = self.L.wf_hip_multi_create(C.byref(cfg), devs, len(devices), streams, ring_frames, C.byref(m))
        if rc != 0:
            raise WfHipError(rc, self.L.wf_hip_multi_last_error(None).decode())
        self.m = m
        self.streams = streams
        self.n_devices = int(self.L.wf_hip_multi_num_devices(m))
        self.transport = self.L.wf_hip_multi_transport(m).decode()
        self.transport_note = self.L.wf_hip_multi_last_error(m).decode()  # why not the transport it would have picked ("" if it did)
        self.shards = []
        for i in range(self.n_devices):
            dev, first, count = C.c_int(0), C.c_uint32(0), C.c_uint32(0)
            h = self.L.wf_hip_multi_shard(m, i, C.byref(dev), C.byref(first), C.byref(count))
            self.shards.append((C.c_void_p(h), int(dev.value), int(first.value), int(count.value)))
        h0 = self.shards[0][0]
        self.fft_size = self.L.wf_hip_fft_size(h0)
        self.bins = self.fft_size // 2
        self.capture_channels = self.L.wf_hip_capture_channels(h0)
        self.output_channels = self.L.wf_hip_output_channels(h0)
        self.display_channels = self.L.wf_hip_display_channels(h0)
        self.num_bars = self.L.wf_hip_num_bars(h0)

    def _ck(self, rc):
        if rc != 0:
            raise WfHipError(rc, self.L.wf_hip_multi_last_error(self.m).decode())

    def close(self):
        if getattr(self, "m", None):
            self.L.wf_hip_multi_destroy(self.m)
            self.m = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def push_audio(self, samples: np.ndarray, first: int = 0):
        s = np.ascontiguousarray(samples, dtype=np.float32)
        assert s.ndim == 3 and s.shape[1] == self.capture_channels, s.shape
        self._ck(self.L.wf_hip_multi_push_audio(self.m, first, s.shape[0], s.ctypes.data_as(C.POINTER(C.c_float)), s.shape[2]))

    def push_synth(self, seed: int, index0: int, frames: int, first: int = 0, count: int | None = None, stream_id0: int = 0):
        count = self.streams - first if count is None else count
        self._ck(self.L.wf_hip_multi_push_synth(self.m, first, count, seed, stream_id0, index0, frames))

    def set_hidden(self, mask, first: int = 0):
        m = np.ascontiguousarray(mask, dtype=np.uint8)
        self._ck(self.L.wf_hip_multi_set_hidden(self.m, first, len(m), m.ctypes.data_as(C.POINTER(C.c_uint8))))

    def reset(self, first: int = 0, count: int | None = None):
        count = self.streams - first if count is None else count
        self._ck(self.L.wf_hip_multi_reset(self.m, first, count))

    def tick(self, seconds: float = 1.0 / 60.0, delay_frames: int = 0, input_rms: float = 0.0, flags: int = 0):
        p = TickParams(seconds, delay_frames, input_rms, flags, 0)
        self._ck(self.L.wf_hip_multi_tick(self.m, C.byref(p)))

    def sync(self):
        self._ck(self.L.wf_hip_multi_sync(self.m))

    def decibels(self, first: int = 0, count: int | None = None) -> np.ndarray:
        count = self.streams - first if count is None else count
        out = np.empty((count, self.output_channels, self.bins), np.float32)
        self._ck(self.L.wf_hip_multi_read(self.m, OUT_DECIBELS, first, count, out.ctypes.data_as(C.c_void_p)))
        return out

    def bars(self, first: int = 0, count: int | None = None) -> np.ndarray:
        count = self.streams - first if count is None else count
        out = np.empty((count, self.display_channels, self.num_bars), np.float32)
        self._ck(self.L.wf_hip_multi_read(self.m, OUT_BARS, first, count, out.ctypes.data_as(C.c_void_p)))
        return out

    def last_silent(self, first: int = 0, count: int | None = None) -> np.ndarray:
        count = self.streams - first if count is None else count
        out = np.empty(count, np.uint8)
        self._ck(self.L.wf_hip_multi_read(self.m, OUT_LAST_SILENT, first, count, out.ctypes.data_as(C.c_void_p)))
        return out.astype(bool)

    def allgather_bars(self):
        """asynchronous: enqueued behind the ticks so far on every device's gather stream"""
        self._ck(self.L.wf_hip_multi_allgather_bars(self.m))

    def gathered(self, device_index: int) -> np.ndarray:
        """device `device_index`'s copy of the newest gathered result: [streams, display_channels, num_bars]"""
        out = np.empty((self.streams, self.display_channels, self.num_bars), np.float32)
        self._ck(self.L.wf_hip_multi_read_gathered(self.m, device_index, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def time_ticks(self, ticks: int, hop: int, first_delay: int, gather: bool = False, seconds: float = 1.0 / 60.0, flags: int = 0):
        """(largest per-device average device ms per tick, [per-device ms])"""
        p = TickParams(seconds, first_delay, 0.0, flags, 0)
        ms = C.c_float(0.0)
        per = (C.c_float * self.n_devices)()
        self._ck(self.L.wf_hip_multi_time_ticks(self.m, C.byref(p), ticks, hop, 1 if gather else 0, C.byref(ms), per))
        return float(ms.value), [float(x) for x in per]

    def algorithmic_bytes_per_tick(self, flags: int = 0) -> int:
        return sum(int(self.L.wf_hip_algorithmic_bytes_per_tick(h, flags)) for h, _, _, _ in self.shards)
