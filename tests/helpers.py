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
        display_mode="level_meter" if cfg.meter else "waveform" if cfg.waveform else
        (("stepped_bars" if getattr(cfg, "vertices", 0) == 3 else "bars") if cfg.bars else "curve"),
        step_width=int(getattr(cfg, "step_width", 8)), step_gap=int(getattr(cfg, "step_gap", 4)),
        rms_mode=bool(cfg.meter_rms), meter_buf=int(cfg.meter_ms), interp_mode=INTERP_KEYS[cfg.interp_mode],
        log_scale=bool(cfg.log_scale), mirror_freq_axis=bool(cfg.mirror_freq_axis),
        width=cfg.width, height=cfg.height, bar_width=cfg.bar_width, bar_gap=cfg.bar_gap,
        channel_spacing=cfg.channel_spacing, min_bar_height=cfg.min_bar_height, rounded_caps=bool(cfg.rounded_caps),
        filter_mode="gauss" if cfg.filter_mode == 1 else "none", filter_radius=repr(float(np.float32(cfg.filter_radius))),
        render_mode="line" if getattr(cfg, "vertices", 0) == 2 else "solid",
    )
    if getattr(cfg, "radial", 0):
        # get_settings halves the height for the radial layout and takes the dead zone off it (src/source.cpp:658-666);
        # wf_config.height is m_height after that
        s.update(radial_layout=True, deadzone=0.0, height=2 * cfg.height)
    s.update(extra)
    return s


# how often the linear arm decided (values that miss the dB arm and pass the linear one), and how many of those lie in the
# range a display or the silence state machine can see (above -75 dB: the default floor - 10) -- reported by
# tests/test_gpu_fuzz.py::test_zz_display_arm_stays_rare so that a regression leaning on the arm shows up as a count
ARM_STATS = {"calls": 0, "values": 0, "linear_arm": 0, "linear_arm_visible": 0, "deep": 0}
VISIBLE_DB = -75.0
# DESIGN.md section 5, deviation (1): the device squares re and im (scaled by 2^40) where the reference calls hypotf; below
# |X| ~ 1e-31 the square underflows and the bin reads DB_MIN (-758.6) where the reference still answers (down to -758).  A value
# the reference puts below DEEP_DB may therefore read lower on the device -- never higher.  Counted ("deep").
DEEP_DB = -600.0

LIN_EPS = 1e-6  # linear-domain arm: |d magnitude| <= LIN_EPS * the largest magnitude of the same frame (row)


def assert_db_close(got, want, what="", lin_eps=LIN_EPS, undo_db=None, deep=False):
    """The parity criterion for rows of dB values (last axis = the bins of one frame), SURVEY.md section 7:

      a value passes if   |got - want| <= RTOL * |want| + ATOL                           (dB arm: 1e-5 relative + 1e-4 dB)
                   or     |10^(got/20) - 10^(want/20)| <= lin_eps * max_k 10^(want_k/20)  (linear arm, per frame)
                   or     deep and want < -600 dB and got <= want                        (the stated floor of the device's |X|, DEEP_DB)

    The second arm is what a float FFT can promise: its error is relative to the level of the whole frame, not to the
    bin -- the reference's own FFTW result misses the dB arm against an exact DFT on bins that sit 60 dB or more under
    their neighbours (a window's DC null, a Rayleigh-distributed noise bin).  lin_eps = 1e-6 is ten times tighter than
    the north star's 1e-5, taken relative to the frame's peak.  undo_db (per bin, e.g. the roll-off table) is added to
    both sides before the linear comparison so that a per-bin attenuation applied after the FFT does not loosen it.
    lin_eps=None: dB arm only (quantities that are not spectra).  deep=True only where `got` is a spectrum row the DEVICE produced
    (the |X|^2 underflow is a property of its kernels): bars, levels, waveform rows and every oracle-vs-reference comparison
    run without the third arm, so a path that wrote DB_MIN into deep bins of any other output fails."""
    got = np.asarray(got, np.float32)
    want = np.asarray(want, np.float32)
    g64, w64 = got.astype(np.float64), want.astype(np.float64)
    err = np.abs(g64 - w64)
    tol = RTOL * np.abs(w64) + ATOL
    bad = err > tol
    if deep and bad.any():
        below = bad & (w64 < DEEP_DB) & (g64 <= w64 + tol)
        ARM_STATS["deep"] += int(below.sum())
        bad &= ~below
    ARM_STATS["calls"] += 1
    ARM_STATS["values"] += int(bad.size)
    if bad.any() and lin_eps and got.ndim >= 1 and got.shape[-1] > 1:
        off = 0.0 if undo_db is None else np.asarray(undo_db, np.float64)
        lg, lw = 10.0 ** ((g64 + off) / 20.0), 10.0 ** ((w64 + off) / 20.0)
        peak = lw.max(axis=-1, keepdims=True)
        saved = bad & (np.abs(lg - lw) <= lin_eps * peak)
        ARM_STATS["linear_arm"] += int(saved.sum())
        ARM_STATS["linear_arm_visible"] += int((saved & (w64 > VISIBLE_DB)).sum())
        bad &= ~saved
    if bad.any():
        i = int(np.argmax(np.where(bad, err - tol, -np.inf)))
        raise AssertionError(f"{what}: {int(bad.sum())} of {bad.size} values off; worst at flat index {i}: "
                             f"got {got.flat[i]!r} want {want.flat[i]!r} (err {err.flat[i]:.3e}, tol {tol.flat[i]:.3e})")
    return float(np.max(err / np.maximum(np.abs(want), 1e-30))) if err.size else 0.0


def assert_levels_close(got, want, exact, what=""):
    """Level-meter values (dBFS).  The reference adds the squares of up to 24000 samples sequentially in float
    (src/source_generic.cpp:236-241): its running total swallows addends below half an ulp and reads up to ~1e-4
    relative LOW; its own AVX variants, which keep eight partial sums, differ from its generic path the same way.
    The device reduces a tree.  A level passes if it is within the dB tolerance of the reference's value (`want`), or
    if it is no farther from the exactly summed value (`exact`: the same restatement with the sum in double) than the
    reference itself is, plus that tolerance -- a path cannot be faulted for being closer to the truth than the
    reference."""
    got = np.asarray(got, np.float32).astype(np.float64)
    want = np.asarray(want, np.float32).astype(np.float64)
    exact = np.asarray(exact, np.float32).astype(np.float64)
    tol = RTOL * np.abs(want) + ATOL
    ok = (np.abs(got - want) <= tol) | (np.abs(got - exact) <= np.abs(want - exact) + tol)
    if not ok.all():
        i = int(np.argmax(~ok))
        raise AssertionError(f"{what}: level {i}: got {got.flat[i]!r}, reference {want.flat[i]!r}, exact sum {exact.flat[i]!r} "
                             f"(tol {tol.flat[i]:.3e})")
