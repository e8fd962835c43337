"""Shared test helpers: map a wf_config onto the reference plugin's own setting keys
(src/settings.hpp) so the same configuration can be given to libwfref (the reference) and to
the library under test."""
from __future__ import annotations

import numpy as np

WINDOW_KEYS = {0: "none", 1: "hann", 2: "hamming", 3: "blackman", 4: "blackman_harris", 5: "power_of_sine"}
TSMOOTH_KEYS = {0: "none", 1: "exp_moving_avg", 2: "tv_exp_moving_avg"}
INTERP_KEYS = {0: "point", 1: "lanczos", 2: "catmull_rom"}

RTOL = 1e-5   # BASELINE.json north_star: 1e-5 relative float tolerance on the outputs
ATOL = 1e-4   # dB; floor for |dB| close to 0 (reference's own generic-vs-AVX paths differ by ~1e-5 dB)


def ref_settings(cfg, **extra) -> dict:
    s = dict(
        fft_size=cfg.fft_size, enable_large_fft=True, auto_fft_size=False,
        channel_mode="stereo" if cfg.stereo else "mono",
        window=WINDOW_KEYS[cfg.window], sine_exponent=cfg.sine_exponent,
        temporal_smoothing=TSMOOTH_KEYS[cfg.tsmoothing], gravity=repr(float(np.float32(cfg.gravity))),
        fast_peaks=bool(cfg.fast_peaks), slope=repr(float(np.float32(cfg.slope))),
        rolloff_q=repr(float(np.float32(cfg.rolloff_q))), rolloff_rate=repr(float(np.float32(cfg.rolloff_rate))),
        cutoff_low=cfg.cutoff_low, cutoff_high=cfg.cutoff_high, floor=cfg.floor_db, ceiling=cfg.ceiling_db,
        normalize_volume=bool(cfg.normalize_volume), volume_target=int(cfg.volume_target), max_gain=int(cfg.max_gain),
        display_mode="level_meter" if cfg.meter else "waveform" if cfg.waveform else ("bars" if cfg.bars else "curve"),
        rms_mode=bool(cfg.meter_rms), meter_buf=int(cfg.meter_ms), interp_mode=INTERP_KEYS[cfg.interp_mode],
        log_scale=bool(cfg.log_scale), mirror_freq_axis=bool(cfg.mirror_freq_axis),
        width=cfg.width, height=cfg.height, bar_width=cfg.bar_width, bar_gap=cfg.bar_gap,
        channel_spacing=cfg.channel_spacing, min_bar_height=cfg.min_bar_height, rounded_caps=bool(cfg.rounded_caps),
        filter_mode="gauss" if cfg.filter_mode == 1 else "none", filter_radius=repr(float(np.float32(cfg.filter_radius))),
    )
    s.update(extra)
    return s


def assert_db_close(got, want, what=""):
    got = np.asarray(got, np.float32)
    want = np.asarray(want, np.float32)
    err = np.abs(got.astype(np.float64) - want.astype(np.float64))
    tol = RTOL * np.abs(want.astype(np.float64)) + ATOL
    bad = err > tol
    if bad.any():
        i = int(np.argmax(err - tol))
        raise AssertionError(f"{what}: {int(bad.sum())} of {bad.size} values off; worst at flat index {i}: "
                             f"got {got.flat[i]!r} want {want.flat[i]!r} (err {err.flat[i]:.3e}, tol {tol.flat[i]:.3e})")
    return float(np.max(err / np.maximum(np.abs(want), 1e-30))) if err.size else 0.0
