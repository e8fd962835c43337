"""ctypes binding of build/libwfemu.so (the wavefront emulator, tests/emu/wf_emu.cpp)."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
import os
LIB_PATH = Path(os.environ["WFEMU_LIBRARY"]) if os.environ.get("WFEMU_LIBRARY") else ROOT / "build" / "libwfemu.so"  # WFEMU_LIBRARY: the sanitizer build
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.environ.get("WFEMU_LIBRARY"):  # (make: a no-op when the library is newer than its sources)
            subprocess.run(["make", "-C", str(ROOT / "tests" / "emu"), str(Path("..") / ".." / "build" / "libwfemu.so")], check=True, capture_output=True)
        L = C.CDLL(str(LIB_PATH))
        fp = C.POINTER(C.c_float)
        L.wfemu_tick.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, fp, C.POINTER(C.c_uint32), C.c_uint32, C.c_float, fp, fp,
                                 C.POINTER(C.c_uint64)]
        L.wfemu_host_table.restype = C.c_long
        L.wfemu_host_table.argtypes = [C.c_void_p, C.c_int, C.c_float, fp, C.c_long]
        L.wfemu_lds_bytes.argtypes = [C.c_uint32]
        _lib = L
    return _lib


def host_table(cfg, which: int, seconds: float = 1 / 60):
    L = lib()
    n = L.wfemu_host_table(C.cast(C.byref(cfg), C.c_void_p), which, seconds, None, 0)
    if n < 0:
        raise ValueError(f"build_host_tables failed: {n}")
    out = np.zeros(max(n, 1), np.float32)
    L.wfemu_host_table(C.cast(C.byref(cfg), C.c_void_p), which, seconds, out.ctypes.data_as(C.POINTER(C.c_float)), n)
    return out[:n]


def bar_lanes(cfg, threads: int, max_blocks: int, which: int):
    """per-thread segment form of the bar tables; None if it does not exist for this configuration"""
    L = lib()
    L.wfemu_bar_lanes.restype = C.c_long
    L.wfemu_bar_lanes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_long]
    n = L.wfemu_bar_lanes(C.cast(C.byref(cfg), C.c_void_p), threads, max_blocks, which, None, 0)
    if n == -1000:
        return None
    if n < 0:
        raise ValueError(f"wfemu_bar_lanes failed: {n}")
    out = np.zeros(max(n, 1), np.float32)
    L.wfemu_bar_lanes(C.cast(C.byref(cfg), C.c_void_p), threads, max_blocks, which, out.ctypes.data_as(C.POINTER(C.c_float)), n)
    return out[:n]


def bar_pieces(cfg, threads: int, points: int, max_blocks: int, which: int):
    """wave-private form of the bar tables (wf::bar_pieces); None if it does not exist for this configuration"""
    L = lib()
    L.wfemu_bar_pieces.restype = C.c_long
    L.wfemu_bar_pieces.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_long]
    n = L.wfemu_bar_pieces(C.cast(C.byref(cfg), C.c_void_p), threads, points, max_blocks, which, None, 0)
    if n == -1000:
        return None
    if n < 0:
        raise ValueError(f"wfemu_bar_pieces failed: {n}")
    out = np.zeros(max(n, 1), np.float32)
    L.wfemu_bar_pieces(C.cast(C.byref(cfg), C.c_void_p), threads, points, max_blocks, which, out.ctypes.data_as(C.POINTER(C.c_float)), n)
    return out[:n]


def bar_ps(cfg, threads: int, which: int):
    """prefix-sum form of the bar tables (wf::bar_ps); None if it does not exist for this configuration"""
    L = lib()
    L.wfemu_bar_ps.restype = C.c_long
    L.wfemu_bar_ps.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_long]
    n = L.wfemu_bar_ps(C.cast(C.byref(cfg), C.c_void_p), threads, which, None, 0)
    if n == -1000:
        return None
    if n < 0:
        raise ValueError(f"wfemu_bar_ps failed: {n}")
    out = np.zeros(max(n, 1), np.float32)
    L.wfemu_bar_ps(C.cast(C.byref(cfg), C.c_void_p), threads, which, out.ctypes.data_as(C.POINTER(C.c_float)), n)
    return out[:n]


def bluestein_rows(np_points: int, c: int):
    """wf::build_bluestein_rows: (L, rowtw [C, R], bhat [L], q [R]) as complex64"""
    L = lib()
    L.wfemu_bluestein_rows.restype = C.c_long
    L.wfemu_bluestein_rows.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_float), C.c_long]
    length = L.wfemu_bluestein_rows(np_points, c, -1, None, 0)
    tabs = []
    for which in range(3):
        n = L.wfemu_bluestein_rows(np_points, c, which, None, 0)
        out = np.zeros(n, np.float32)
        L.wfemu_bluestein_rows(np_points, c, which, out.ctypes.data_as(C.POINTER(C.c_float)), n)
        tabs.append(out.view(np.complex64))
    return int(length), tabs[0].reshape(c, np_points // c), tabs[1], tabs[2]


def tick(cfg, ring: np.ndarray, wpos: int, tsmooth: np.ndarray, delay: int = 0, seconds: float = 1 / 60):
    """ring: float32 [spectra, ring_cap]; tsmooth: float32 [spectra, M] (updated in place).
    Returns (decibels [streams, out_ch, M], stats[6])"""
    L = lib()
    n_spec, cap = ring.shape
    streams = n_spec // cfg.capture_channels
    out_ch = 2 if (cfg.capture_channels > 1 or cfg.stereo) else 1
    M = cfg.fft_size // 2
    db = np.zeros((streams, out_ch, M), np.float32)
    w = (C.c_uint32 * streams)(*([wpos] * streams))
    stats = (C.c_uint64 * 6)()
    fp = C.POINTER(C.c_float)
    rc = L.wfemu_tick(C.cast(C.byref(cfg), C.c_void_p), streams, cap, ring.ctypes.data_as(fp), w, delay, seconds,
                      tsmooth.ctypes.data_as(fp), db.ctypes.data_as(fp), stats)
    if rc != 0:
        raise RuntimeError(f"wfemu_tick rc={rc}")
    return db, list(stats)
