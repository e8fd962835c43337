"""ctypes binding of oracle/_ref/libwfref.so -- TEST INFRASTRUCTURE.

libwfref.so is the reference itself (phandasm/waveform v1.9.1 compiled verbatim
against a headless fake libobs, see oracle/ref/wfref.h).  Only tests/,
tools/make_golden.py, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this module; the product package never does.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
# WFREF_LIBRARY: another build of the harness -- oracle/_ref/libwfref_plugin.so is the reference with
# integration/waveform-hip.patch applied (sources created through its own callbacks::create)
LIB_PATH = Path(os.environ["WFREF_LIBRARY"]) if os.environ.get("WFREF_LIBRARY") else _HERE / "_ref" / "libwfref.so"

_lib = None


def available() -> bool:
    return LIB_PATH.exists()


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise FileNotFoundError(
            f"{LIB_PATH} missing: build it with `make -C oracle/ref` where /root/reference exists")
    L = C.CDLL(str(LIB_PATH))
    vp, f32p, u64, u32 = C.c_void_p, C.POINTER(C.c_float), C.c_uint64, C.c_uint32
    L.wfref_create.restype = vp
    L.wfref_create.argtypes = [C.c_char_p, C.c_char_p, u32, C.c_int, u32, u32]
    L.wfref_destroy.argtypes = [vp]
    L.wfref_update.argtypes = [vp, C.c_char_p]
    L.wfref_set_clock_ns.argtypes = [u64]
    L.wfref_clock_ns.restype = u64
    L.wfref_push_audio.argtypes = [vp, f32p, f32p, u32, u64, C.c_int]
    L.wfref_feed_and_tick.argtypes = [vp, f32p, f32p, u32, u64, C.c_float]
    L.wfref_tick.argtypes = [vp, C.c_float]
    L.wfref_render.argtypes = [vp]
    L.wfref_show.argtypes = [vp, C.c_int]
    L.wfref_fft_size.restype = C.c_size_t
    L.wfref_fft_size.argtypes = [vp]
    for name in ("wfref_capture_channels", "wfref_output_channels"):
        getattr(L, name).restype = u32
        getattr(L, name).argtypes = [vp]
    for name in ("wfref_stereo", "wfref_last_silent", "wfref_num_bars", "wfref_using_hip"):
        getattr(L, name).restype = C.c_int
        getattr(L, name).argtypes = [vp]
    L.wfref_class_name.restype = C.c_char_p
    L.wfref_class_name.argtypes = [vp]
    L.wfref_hip_fallback_ticks.restype = C.c_uint64
    L.wfref_hip_fallback_ticks.argtypes = []
    L.wfref_hip_host_rms_updates.restype = C.c_uint64
    L.wfref_hip_host_rms_updates.argtypes = []
    for n in ("wfref_hip_device_renders", "wfref_hip_host_renders"):
        getattr(L, n).restype = C.c_uint64
        getattr(L, n).argtypes = []
    L.wfref_meter_mode.restype = C.c_int
    L.wfref_meter_mode.argtypes = [vp]
    for name in ("wfref_meter_val", "wfref_meter_buf"):
        getattr(L, name).restype = C.c_float
        getattr(L, name).argtypes = [vp, C.c_int]
    L.wfref_input_rms.restype = C.c_float
    L.wfref_input_rms.argtypes = [vp]
    L.wfref_decibels_size.restype = C.c_size_t
    L.wfref_decibels_size.argtypes = [vp]
    L.wfref_ring_bytes.restype = C.c_size_t
    L.wfref_ring_bytes.argtypes = [vp, C.c_int]
    L.wfref_gravity.restype = C.c_float
    L.wfref_gravity.argtypes = [vp, C.c_float]
    L.wfref_db_min.restype = C.c_float
    L.wfref_window_sum.restype = C.c_float
    L.wfref_window_sum.argtypes = [vp]
    for name in ("wfref_decibels", "wfref_tsmooth"):
        getattr(L, name).restype = f32p
        getattr(L, name).argtypes = [vp, C.c_int]
    for name in ("wfref_window", "wfref_slope", "wfref_rolloff"):
        getattr(L, name).restype = f32p
        getattr(L, name).argtypes = [vp]
    L.wfref_interp_indices.restype = C.c_size_t
    L.wfref_interp_indices.argtypes = [vp, C.POINTER(f32p)]
    L.wfref_band_widths.restype = C.c_size_t
    L.wfref_band_widths.argtypes = [vp, C.POINTER(C.POINTER(C.c_int))]
    L.wfref_interp_kernel.restype = C.c_size_t
    L.wfref_interp_kernel.argtypes = [vp, C.POINTER(f32p), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.wfref_bars.restype = C.c_size_t
    L.wfref_bars.argtypes = [vp, C.c_int, C.POINTER(f32p)]
    L.wfref_bench.restype = C.c_double
    L.wfref_bench.argtypes = [C.c_char_p, C.c_char_p, u32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                              u64, C.POINTER(C.c_double)]
    L.wfref_noise.restype = C.c_float
    L.wfref_noise.argtypes = [u64, u32, u32, u64]
    _lib = L
    return L


def _arr(ptr, n, dtype=np.float32):
    if not ptr or n == 0:
        return None
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


def settings_str(settings: dict | None) -> bytes:
    if not settings:
        return b""
    parts = []
    for k, v in settings.items():
        if isinstance(v, bool):
            v = "true" if v else "false"
        parts.append(f"{k}={v}")
    return ";".join(parts).encode()


class RefSource:
    """One reference WAVSource driven the way OBS drives it."""

    def __init__(self, settings: dict | None = None, isa: str = "generic", sample_rate: int = 48000,
                 channels: int = 2, fps=(60, 1)):
        self.L = lib()
        self.sample_rate = sample_rate
        self.h = self.L.wfref_create(isa.encode(), settings_str(settings), sample_rate, channels, fps[0], fps[1])
        if not self.h:
            raise RuntimeError("wfref_create failed")
        self.now_ns = 1_000_000_000

    def close(self):
        if self.h:
            self.L.wfref_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def update(self, settings: dict):
        self.L.wfref_update(self.h, settings_str(settings))

    # -- driving ---------------------------------------------------------
    def feed_and_tick(self, ch0, ch1=None, seconds: float = 1.0 / 60.0, advance_ns: int | None = None):
        """Push len(ch0) new frames per channel ending 'now', then tick."""
        ch0 = np.ascontiguousarray(ch0, dtype=np.float32)
        n = len(ch0)
        p0 = ch0.ctypes.data_as(C.POINTER(C.c_float))
        if ch1 is not None:
            ch1 = np.ascontiguousarray(ch1, dtype=np.float32)
            p1 = ch1.ctypes.data_as(C.POINTER(C.c_float))
        else:
            p1 = C.POINTER(C.c_float)()
        if advance_ns is None:
            advance_ns = n * 1_000_000_000 // self.sample_rate
        self.now_ns += advance_ns
        self.L.wfref_feed_and_tick(self.h, p0, p1, n, self.now_ns, seconds)

    def tick_only(self, seconds: float = 1.0 / 60.0, advance_ns: int = 16_666_667):
        self.now_ns += advance_ns
        self.L.wfref_set_clock_ns(self.now_ns)
        self.L.wfref_tick(self.h, seconds)

    def render(self):
        self.L.wfref_render(self.h)

    def show(self, flag: bool):
        self.L.wfref_show(self.h, 1 if flag else 0)

    # -- state -------------------------------------------------------------
    @property
    def fft_size(self):
        return self.L.wfref_fft_size(self.h)

    @property
    def capture_channels(self):
        return self.L.wfref_capture_channels(self.h)

    @property
    def output_channels(self):
        return self.L.wfref_output_channels(self.h)

    @property
    def stereo(self):
        return bool(self.L.wfref_stereo(self.h))

    @property
    def using_hip(self):
        return bool(self.L.wfref_using_hip(self.h))

    @property
    def class_name(self) -> str:
        """"hip" / "avx2" / "avx" / "generic": the WAVSource subclass behind this source"""
        return self.L.wfref_class_name(self.h).decode()

    @property
    def last_silent(self):
        return bool(self.L.wfref_last_silent(self.h))

    @property
    def num_bars(self):
        return self.L.wfref_num_bars(self.h)

    @property
    def meter_mode(self):
        return bool(self.L.wfref_meter_mode(self.h))

    def meter_val(self, ch):
        return float(self.L.wfref_meter_val(self.h, ch))

    def meter_buf(self, ch):
        return float(self.L.wfref_meter_buf(self.h, ch))

    @property
    def input_rms(self):
        return float(self.L.wfref_input_rms(self.h))

    def ring_bytes(self, ch):
        return self.L.wfref_ring_bytes(self.h, ch)

    def gravity(self, seconds):
        return float(self.L.wfref_gravity(self.h, seconds))

    def decibels(self, ch):
        return _arr(self.L.wfref_decibels(self.h, ch), self.L.wfref_decibels_size(self.h))

    def tsmooth(self, ch):
        return _arr(self.L.wfref_tsmooth(self.h, ch), self.fft_size // 2)

    def window(self):
        return _arr(self.L.wfref_window(self.h), self.fft_size)

    def window_sum(self):
        return float(self.L.wfref_window_sum(self.h))

    def slope(self):
        return _arr(self.L.wfref_slope(self.h), self.fft_size // 2)

    def rolloff(self):
        return _arr(self.L.wfref_rolloff(self.h), self.fft_size // 2)

    def interp_indices(self):
        p = C.POINTER(C.c_float)()
        n = self.L.wfref_interp_indices(self.h, C.byref(p))
        return _arr(p, n)

    def band_widths(self):
        p = C.POINTER(C.c_int)()
        n = self.L.wfref_band_widths(self.h, C.byref(p))
        return _arr(p, n, np.int32)

    def interp_kernel(self):
        p = C.POINTER(C.c_float)()
        r, s = C.c_int(0), C.c_int(0)
        n = self.L.wfref_interp_kernel(self.h, C.byref(p), C.byref(r), C.byref(s))
        return _arr(p, n), r.value, s.value

    def bars(self, ch):
        p = C.POINTER(C.c_float)()
        n = self.L.wfref_bars(self.h, ch, C.byref(p))
        return _arr(p, n)

    def shader_value(self, name: str):
        """the last value set_shader_vars (src/source.cpp:1693-1770) gave shader parameter `name` on this thread: float32[4]
        (a float parameter in [0]), or None if it was never set"""
        self.L.wfref_shader_value.restype = C.c_int
        self.L.wfref_shader_value.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_float)]
        out = (C.c_float * 4)()
        if not self.L.wfref_shader_value(self.h, name.encode(), out):
            return None
        return np.array(list(out), np.float32)

    def draws(self):
        """the gs_draw calls of the last render(): [(mode, vertices [n, 4])], one per displayed channel"""
        self.L.wfref_draw_count.restype = C.c_int
        self.L.wfref_draw_count.argtypes = [C.c_void_p]
        self.L.wfref_draw.restype = C.c_size_t
        self.L.wfref_draw.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_float))]
        out = []
        for i in range(self.L.wfref_draw_count(self.h)):
            mode, p = C.c_int(0), C.POINTER(C.c_float)()
            n = self.L.wfref_draw(self.h, i, C.byref(mode), C.byref(p))
            out.append((mode.value, _arr(p, n * 4).reshape(n, 4).copy() if n else np.zeros((0, 4), np.float32)))
        return out


def thread_stress(isa: str, settings: dict | None, n_sources: int, seconds: float, seed: int = 7):
    """(sources that differ from a fresh one after the run, stats dict) -- wfref_thread_stress (oracle/ref/wfref.h)"""
    L = lib()
    L.wfref_thread_stress.restype = C.c_int
    L.wfref_thread_stress.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_double, C.c_uint64, C.POINTER(C.c_uint64)]
    st = (C.c_uint64 * 6)()
    rc = L.wfref_thread_stress(isa.encode(), settings_str(settings), n_sources, seconds, seed, st)
    return rc, dict(zip(("ticks", "packets", "updates", "recreations", "renders", "sources"), (int(v) for v in st)))


def hip_device_renders() -> int:
    """render() calls of WAVSourceHIP spectrum sources drawn from the device's vertices"""
    return int(lib().wfref_hip_device_renders())


def hip_host_renders() -> int:
    """render() calls of WAVSourceHIP spectrum sources that ran the reference's own interpolation and vertex loops"""
    return int(lib().wfref_hip_host_renders())


def hip_host_rms_updates() -> int:
    """update_input_rms calls of WAVSourceHIP sources that ran the reference's host loop"""
    return int(lib().wfref_hip_host_rms_updates())


def hip_fallback_ticks() -> int:
    """ticks WAVSourceHIP handed to the reference's CPU class so far (process-wide)"""
    return int(lib().wfref_hip_fallback_ticks())


def db_min() -> float:
    return float(lib().wfref_db_min())


def bench(isa: str, settings: dict | None, n_streams: int, n_threads: int, warmup: int, ticks: int, hop: int = 800,
          sample_rate: int = 48000, channels: int = 2, seed: int = 0x5741564546524D31, render: bool = False):
    """Returns (spectra_per_s, elapsed_s) for the reference CPU path.  render: every frame also renders every source."""
    el = C.c_double(0.0)
    lib().wfref_bench_set_render(1 if render else 0)
    v = lib().wfref_bench(isa.encode(), settings_str(settings), sample_rate, channels, n_streams, n_threads, warmup,
                          ticks, hop, seed, C.byref(el))
    return float(v), float(el.value)
