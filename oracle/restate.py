"""ctypes binding of oracle/libwforacle.so -- the CPU restatement of the reference path.
TEST INFRASTRUCTURE: only tests/, tools/make_golden.py, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "libwforacle.so"
_lib = None


def available() -> bool:
    return LIB_PATH.exists()


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise FileNotFoundError(f"{LIB_PATH} missing: run `make -C oracle`")
    L = C.CDLL(str(LIB_PATH))
    vp, fp = C.c_void_p, C.POINTER(C.c_float)
    L.wfo_create.restype = vp
    L.wfo_create.argtypes = [vp]
    L.wfo_destroy.argtypes = [vp]
    L.wfo_push_audio.argtypes = [vp, fp, fp, C.c_uint32, C.c_int]
    L.wfo_set_sync_delay.argtypes = [vp, C.c_uint32]
    L.wfo_set_hidden.argtypes = [vp, C.c_int]
    L.wfo_set_input_rms.argtypes = [vp, C.c_float]
    L.wfo_tick.argtypes = [vp, C.c_float]
    L.wfo_update_input_rms.argtypes = [vp]
    L.wfo_input_rms.restype = C.c_float
    L.wfo_input_rms.argtypes = [vp]
    L.wfo_render_bars.argtypes = [vp]
    L.wfo_fft_size.restype = C.c_uint32
    L.wfo_fft_size.argtypes = [vp]
    L.wfo_output_channels.restype = C.c_uint32
    L.wfo_output_channels.argtypes = [vp]
    L.wfo_last_silent.restype = C.c_int
    L.wfo_last_silent.argtypes = [vp]
    L.wfo_ring_samples.restype = C.c_size_t
    L.wfo_ring_samples.argtypes = [vp, C.c_int]
    L.wfo_gravity.restype = C.c_float
    L.wfo_gravity.argtypes = [vp, C.c_float]
    L.wfo_db_min.restype = C.c_float
    for n in ("wfo_decibels", "wfo_tsmooth", "wfo_tsmooth_mut", "wfo_bars"):
        getattr(L, n).restype = fp
        getattr(L, n).argtypes = [vp, C.c_int]
    L.wfo_window.restype = fp
    L.wfo_window.argtypes = [vp, fp]
    for n in ("wfo_slope", "wfo_rolloff"):
        getattr(L, n).restype = fp
        getattr(L, n).argtypes = [vp]
    L.wfo_num_bars.restype = C.c_int
    L.wfo_num_bars.argtypes = [vp]
    L.wfo_interp_indices.restype = C.c_size_t
    L.wfo_interp_indices.argtypes = [vp, C.POINTER(fp)]
    L.wfo_band_widths.restype = C.c_size_t
    L.wfo_band_widths.argtypes = [vp, C.POINTER(C.POINTER(C.c_int))]
    L.wfo_interp_weights.restype = C.c_size_t
    L.wfo_interp_weights.argtypes = [vp, C.POINTER(fp), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.wfo_r2c.argtypes = [fp, C.c_uint32, fp]
    L.wfo_wave_create.restype = vp
    L.wfo_wave_create.argtypes = [vp]
    L.wfo_wave_destroy.argtypes = [vp]
    L.wfo_wave_push_audio.argtypes = [vp, fp, fp, C.c_uint32, C.c_int]
    L.wfo_wave_set_time.argtypes = [vp, C.c_uint64, C.c_uint32]
    L.wfo_wave_set_hidden.argtypes = [vp, C.c_int]
    L.wfo_wave_set_input_rms.argtypes = [vp, C.c_float]
    L.wfo_wave_tick.argtypes = [vp]
    L.wfo_wave_update_input_rms.restype = C.c_float
    L.wfo_wave_update_input_rms.argtypes = [vp]
    for n in ("wfo_wave_points", "wfo_wave_output_channels"):
        getattr(L, n).restype = C.c_uint32
        getattr(L, n).argtypes = [vp]
    L.wfo_wave_last_silent.restype = C.c_int
    L.wfo_wave_last_silent.argtypes = [vp]
    L.wfo_wave_row.restype = fp
    L.wfo_wave_row.argtypes = [vp, C.c_int]
    L.wfo_wave_ts.restype = C.c_uint64
    L.wfo_wave_ts.argtypes = [vp]
    L.wfo_meter_create.restype = vp
    L.wfo_meter_create.argtypes = [vp]
    L.wfo_meter_destroy.argtypes = [vp]
    L.wfo_meter_push_audio.argtypes = [vp, fp, fp, C.c_uint32, C.c_int]
    L.wfo_meter_set_sync_delay.argtypes = [vp, C.c_uint32]
    L.wfo_meter_set_state.argtypes = [vp, C.c_int]
    L.wfo_meter_set_exact.argtypes = [vp, C.c_int]
    L.wfo_meter_tick.argtypes = [vp, C.c_float]
    L.wfo_meter_render.argtypes = [vp]
    L.wfo_meter_size.restype = C.c_uint32
    L.wfo_meter_size.argtypes = [vp]
    L.wfo_meter_last_silent.restype = C.c_int
    L.wfo_meter_last_silent.argtypes = [vp]
    for n in ("wfo_meter_val", "wfo_meter_ema", "wfo_meter_bar"):
        getattr(L, n).restype = C.c_float
        getattr(L, n).argtypes = [vp, C.c_int]
    _lib = L
    return L


def _arr(ptr, n, dtype=np.float32):
    if not ptr or n == 0:
        return None
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


def db_min() -> float:
    return float(lib().wfo_db_min())


def r2c(x: np.ndarray) -> np.ndarray:
    """complex64[n/2]: the DFT stage alone (double-precision DFT rounded to float)"""
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty(len(x), np.float32)
    lib().wfo_r2c(x.ctypes.data_as(C.POINTER(C.c_float)), len(x), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out.view(np.complex64)


class OracleSource:
    """One restated WAVSource.  `cfg` is any ctypes struct laid out as wf_config (include/wf_config.h)."""

    def __init__(self, cfg):
        self.L = lib()
        self._cfg = cfg
        self.h = self.L.wfo_create(C.cast(C.byref(cfg), C.c_void_p))
        if not self.h:
            raise ValueError("wfo_create rejected the configuration")
        self.fft_size = self.L.wfo_fft_size(self.h)
        self.bins = self.fft_size // 2
        self.capture_channels = int(cfg.capture_channels)
        self.output_channels = self.L.wfo_output_channels(self.h)
        self.display_channels = 2 if cfg.stereo else 1
        self.num_bars = self.L.wfo_num_bars(self.h)

    def close(self):
        if getattr(self, "h", None):
            self.L.wfo_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def push_audio(self, audio, muted=False):
        """audio: float32 [capture_channels, frames]"""
        a = np.ascontiguousarray(audio, np.float32)
        assert a.ndim == 2 and a.shape[0] >= self.capture_channels
        fp = C.POINTER(C.c_float)
        p0 = a[0].ctypes.data_as(fp)
        p1 = a[1].ctypes.data_as(fp) if self.capture_channels > 1 else fp()
        self.L.wfo_push_audio(self.h, p0, p1, a.shape[1], 1 if muted else 0)

    def set_sync_delay(self, frames):
        self.L.wfo_set_sync_delay(self.h, frames)

    def set_hidden(self, hidden):
        self.L.wfo_set_hidden(self.h, 1 if hidden else 0)

    def set_input_rms(self, rms):
        self.L.wfo_set_input_rms(self.h, rms)

    def update_input_rms(self):
        """update_input_rms() from the captured audio (cfg.normalize_volume); returns m_input_rms"""
        self.L.wfo_update_input_rms(self.h)
        return float(self.L.wfo_input_rms(self.h))

    def tick(self, seconds=1.0 / 60.0):
        self.L.wfo_tick(self.h, seconds)

    def feed_and_tick(self, audio, seconds=1.0 / 60.0):
        self.push_audio(audio)
        self.tick(seconds)

    def render_bars(self):
        self.L.wfo_render_bars(self.h)

    def vertices(self, channel, line=False):
        """the vertices render_bars / render_curve write for one displayed channel (after render_bars()): [n, 4]"""
        self.L.wfo_fill_vertices.restype = C.c_size_t
        self.L.wfo_fill_vertices.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_size_t]
        n = self.L.wfo_fill_vertices(self.h, channel, 1 if line else 0, None, 0)
        out = np.zeros((max(n, 1), 4), np.float32)
        self.L.wfo_fill_vertices(self.h, channel, 1 if line else 0, out.ctypes.data_as(C.POINTER(C.c_float)), n)
        return out[:n]

    @property
    def last_silent(self):
        return bool(self.L.wfo_last_silent(self.h))

    def ring_samples(self, ch):
        return self.L.wfo_ring_samples(self.h, ch)

    def gravity(self, seconds):
        return float(self.L.wfo_gravity(self.h, seconds))

    def decibels(self, ch=None):
        """[display_channels, bins] (or one channel)"""
        if ch is not None:
            return _arr(self.L.wfo_decibels(self.h, ch), self.bins)
        return np.stack([_arr(self.L.wfo_decibels(self.h, c), self.bins) for c in range(self.display_channels)])

    def tsmooth(self, ch):
        return _arr(self.L.wfo_tsmooth(self.h, ch), self.bins)

    def window(self):
        s = C.c_float(0)
        p = self.L.wfo_window(self.h, C.byref(s))
        return _arr(p, self.fft_size), float(s.value)

    def slope(self):
        return _arr(self.L.wfo_slope(self.h), self.bins)

    def rolloff(self):
        return _arr(self.L.wfo_rolloff(self.h), self.bins)

    def interp_indices(self):
        p = C.POINTER(C.c_float)()
        n = self.L.wfo_interp_indices(self.h, C.byref(p))
        return _arr(p, n)

    def band_widths(self):
        p = C.POINTER(C.c_int)()
        n = self.L.wfo_band_widths(self.h, C.byref(p))
        return _arr(p, n, np.int32)

    def interp_weights(self):
        p, r, t = C.POINTER(C.c_float)(), C.c_int(0), C.c_int(0)
        n = self.L.wfo_interp_weights(self.h, C.byref(p), C.byref(r), C.byref(t))
        return _arr(p, n), r.value, t.value

    def bars(self, ch=None):
        if self.num_bars == 0:  # a display narrower than one bar: m_num_bars == 0, nothing is drawn
            return None
        if ch is not None:
            return _arr(self.L.wfo_bars(self.h, ch), self.num_bars)
        return np.stack([_arr(self.L.wfo_bars(self.h, c), self.num_bars) for c in range(self.display_channels)])


class OracleMeter:
    """One restated WAVSource in level-meter display mode (oracle/wf_oracle_meter.c)."""

    def __init__(self, cfg):
        self.L = lib()
        self._cfg = cfg
        self.h = self.L.wfo_meter_create(C.cast(C.byref(cfg), C.c_void_p))
        if not self.h:
            raise ValueError("wfo_meter_create rejected the configuration")
        self.size = self.L.wfo_meter_size(self.h)
        self.capture_channels = int(cfg.capture_channels)

    def close(self):
        if getattr(self, "h", None):
            self.L.wfo_meter_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def push_audio(self, audio, muted=False):
        a = np.ascontiguousarray(audio, np.float32)
        fp = C.POINTER(C.c_float)
        p0 = a[0].ctypes.data_as(fp)
        p1 = a[1].ctypes.data_as(fp) if self.capture_channels > 1 else fp()
        self.L.wfo_meter_push_audio(self.h, p0, p1, a.shape[1], 1 if muted else 0)

    def set_sync_delay(self, frames):
        self.L.wfo_meter_set_sync_delay(self.h, frames)

    def set_state(self, state):
        """0 shown, 1 hidden (!m_show), 2 capture timed out"""
        self.L.wfo_meter_set_state(self.h, state)

    def set_exact(self, exact=True):
        """RMS sums accumulated in double: the value the reference's sequential float sum approximates"""
        self.L.wfo_meter_set_exact(self.h, 1 if exact else 0)

    def tick(self, seconds=1.0 / 60.0):
        self.L.wfo_meter_tick(self.h, seconds)

    @property
    def last_silent(self):
        return bool(self.L.wfo_meter_last_silent(self.h))

    def levels(self):
        return np.array([self.L.wfo_meter_val(self.h, c) for c in range(self.capture_channels)], np.float32)

    def ema(self):
        return np.array([self.L.wfo_meter_ema(self.h, c) for c in range(self.capture_channels)], np.float32)

    def bars(self):
        self.L.wfo_meter_render(self.h)
        return np.array([self.L.wfo_meter_bar(self.h, c) for c in range(self.capture_channels)], np.float32)


class OracleWave:
    """One restated WAVSource in waveform display mode (oracle/wf_oracle_wave.c)."""

    def __init__(self, cfg):
        self.L = lib()
        self._cfg = cfg
        self.h = self.L.wfo_wave_create(C.cast(C.byref(cfg), C.c_void_p))
        if not self.h:
            raise ValueError("wfo_wave_create rejected the configuration")
        self.points = self.L.wfo_wave_points(self.h)
        self.capture_channels = int(cfg.capture_channels)
        self.output_channels = self.L.wfo_wave_output_channels(self.h)
        self.display_channels = 2 if cfg.stereo else 1

    def close(self):
        if getattr(self, "h", None):
            self.L.wfo_wave_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_time(self, audio_ts_ns, reserve_frames=0):
        self.L.wfo_wave_set_time(self.h, audio_ts_ns, reserve_frames)

    def push_audio(self, audio, muted=False):
        a = np.ascontiguousarray(audio, np.float32)
        fp = C.POINTER(C.c_float)
        p0 = a[0].ctypes.data_as(fp)
        p1 = a[1].ctypes.data_as(fp) if self.capture_channels > 1 else fp()
        self.L.wfo_wave_push_audio(self.h, p0, p1, a.shape[1], 1 if muted else 0)

    def set_hidden(self, hidden):
        self.L.wfo_wave_set_hidden(self.h, 1 if hidden else 0)

    def set_input_rms(self, rms):
        self.L.wfo_wave_set_input_rms(self.h, rms)

    def update_input_rms(self):
        return float(self.L.wfo_wave_update_input_rms(self.h))

    def tick(self):
        self.L.wfo_wave_tick(self.h)

    @property
    def last_silent(self):
        return bool(self.L.wfo_wave_last_silent(self.h))

    @property
    def waveform_ts(self):
        return int(self.L.wfo_wave_ts(self.h))

    def rows(self):
        """[display_channels, points]"""
        return np.stack([_arr(self.L.wfo_wave_row(self.h, c), self.points) for c in range(self.display_channels)])
