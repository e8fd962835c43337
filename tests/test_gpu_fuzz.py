"""Randomised parity: libwaveform_hip.so against the CPU restatement (oracle/wf_oracle.c, itself pinned to the
reference by tests/test_golden.py) over configurations and event sequences nobody wrote by hand.

Every case is a seeded draw of
  * a configuration: FFT size, channel layout (mono, mono mixdown, stereo, one channel shown twice), window, smoothing
    mode / gravity / fast peaks, slope, roll-off, volume normalisation, display (none, bars, curve), interpolation,
    log/linear axis, mirror, geometry, Gaussian filter;
  * a script: noise packets of ragged sizes, digital silence long enough to reach m_last_silent, a half-silent
    stretch, muted packets, hide/show, ticks with varying frame times.
Same tolerances as the golden tests.  The draws are deterministic (seeded), so a failure names its case.
"""
import numpy as np
import pytest

import scenarios
from helpers import assert_db_close, assert_levels_close

# Consecutive seeds, nothing picked: every family runs range(N) of its own draw function.
SPEC_SEEDS = range(600)    # power-of-two FFT sizes 128 .. MAX_POW2
BLU_SEEDS = range(500)     # every other multiple of 16 (Bluestein)
REF_EVERY = 1              # every spectrum seed is also played against libwfref.so (the reference itself, float FFTW)
MAX_POW2 = 32768
MAX_ANY = 10912
RESTATEMENT_MAX_PRIME = 61  # lengths whose largest prime factor exceeds this are checked against libwfref.so only
SMOOTH_SEEDS = range(200)  # fft sizes with no prime factor above 13 that are not powers of two: the mixed-radix path (wf_mixed.hpp)
HUGE_SEEDS = range(60)     # the sizes beyond a CU's LDS (wf_big.hpp): 65536 and every other multiple of 16 above 10912


def _largest_prime_factor(n: int) -> int:
    p, best = 2, 1
    while p * p <= n:
        while n % p == 0:
            best, n = p, n // p
        p += 1
    return max(best, n) if n > 1 else best


def _five_smooth(n):  # (what plan_mixed_radix takes: no prime factor above 13, one prime factor of 17 .. 127 on top at most)
    for p in (17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83, 89, 97, 101, 103, 107, 109, 113, 127):
        if n % p == 0:
            n //= p
            break
    for p in (2, 3, 5, 7, 11, 13):
        while n % p == 0:
            n //= p
    return n == 1


SMOOTH_SIZES = [n for n in range(128, 16384 + 1, 16) if _five_smooth(n) and n & (n - 1)]
HUGE_SMOOTH_SIZES = [n for n in range(16384 + 16, 65536, 16) if _five_smooth(n) and n & (n - 1)]


def draw(seed: int, family: str = "pow2", fft_size: int | None = None):
    r = np.random.default_rng({"pow2": 1000, "any": 77000, "huge": 555000, "smooth": 880000}[family] + seed)
    if family == "huge":
        # 65536 itself a quarter of the time, else ANY multiple of 16 in (10912, 65536) -- awkward prime factors included: that
        # is where the device's Bluestein path matters.  The restatement's DFT of a length with a large prime factor p costs
        # O(n p) in double per channel and tick, so run_spectrum_case checks such lengths against libwfref.so (the reference's
        # own FFTW, fast at every length) alone and plays the restatement only where p <= RESTATEMENT_MAX_PRIME
        u = r.random()
        if u < 0.25:
            n = 65536
        elif u < 0.5:  # the sizes above 16384 with small prime factors: rows of a mixed-radix transform (big_mr_rows_kernel)
            n = int(r.choice(HUGE_SMOOTH_SIZES))
        else:
            while True:
                n = 16 * int(r.integers(10912 // 16 + 1, 65536 // 16))
                if n & (n - 1):
                    break
    elif family == "smooth":
        # the sizes the mixed-radix path takes (wf_mixed.hpp): multiples of 16 up to 16384 with no prime factor above 13 that
        # are not powers of two -- the automatic sizes (800, 1600, 960, 1920, 2000 at 48 kHz; 1760, 1456, 880 at 44.1 kHz) a fifth of the time
        if r.random() < 0.2:
            n = int(r.choice([800, 1600, 960, 1920, 2000, 400, 320, 1760, 1456, 880]))
        else:
            n = int(r.choice(SMOOTH_SIZES))
    elif family == "pow2":
        sizes = [128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536]
        p = np.array([0.07, 0.07, 0.08, 0.2, 0.17, 0.17, 0.09, 0.07, 0.04, 0.04])
        keep = [i for i, v in enumerate(sizes) if v <= MAX_POW2]
        n = int(r.choice([sizes[i] for i in keep], p=p[keep] / p[keep].sum()))
    else:  # any multiple of 16 in [128, MAX_ANY] that is not a power of two; small sizes as likely as large ones
        hi = float(r.choice([1024, 4096, MAX_ANY]))
        n = 16 * int(r.integers(8, int(hi) // 16 + 1))
        if n & (n - 1) == 0:
            n += 16
    if fft_size is not None:  # (tests/sizes_large_sweep.py: everything else as drawn)
        n = int(fft_size)
    layout = int(r.integers(0, 4))  # 0 mono capture, 1 mono mixdown of 2, 2 stereo, 3 one captured channel shown twice
    cfg = dict(fft_size=n,
               capture_channels=1 if layout in (0, 3) else 2,
               stereo=1 if layout in (2, 3) else 0,
               window=int(r.integers(0, 6)), sine_exponent=int(r.integers(1, 5)),
               tsmoothing=int(r.integers(0, 3)), gravity=float(np.float32(r.uniform(0.05, 0.95))),
               fast_peaks=int(r.integers(0, 2)),
               slope=float(np.float32(r.choice([0.0, 0.5, 1.0, 2.5]))),
               floor_db=int(r.choice([-65, -80, -50])), ceiling_db=int(r.choice([0, -6])))
    if r.random() < 0.35:
        cfg.update(rolloff_q=float(np.float32(r.uniform(0.5, 3.0))), rolloff_rate=float(np.float32(r.uniform(3.0, 24.0))))
    if r.random() < 0.3:
        cfg.update(normalize_volume=1, volume_target=float(r.integers(-20, -2)), max_gain=float(r.integers(6, 31)))  # integer sliders in the reference
    display = int(r.integers(0, 3))  # 0 spectrum only, 1 bars, 2 curve
    if display:
        cfg.update(interp_mode=int(r.integers(0, 3)), log_scale=int(r.integers(0, 2)), mirror_freq_axis=int(r.random() < 0.3),
                   height=int(r.choice([225, 300, 101])), channel_spacing=int(r.choice([0, 0, 6])))
        if display == 1:
            cfg.update(bars=1, width=int(r.choice([800, 640, 1000])), bar_width=int(r.choice([24, 12, 5])), bar_gap=int(r.choice([6, 2, 0])),
                       min_bar_height=int(r.choice([0, 3])), rounded_caps=int(r.random() < 0.3))
        else:
            cfg.update(curve=1, width=int(r.choice([800, 500, 333, 1024])))
        if r.random() < 0.5:
            cfg.update(filter_mode=1, filter_radius=float(np.float32(r.choice([0.4, 1.5, 3.0, 7.5]))))
    steps = []
    for _ in range(int(r.integers(3, 6))):
        steps += [("noise", int(r.choice([800, 441, 1024, 37, 1600]))), ("tick", float(np.float32(r.choice([1 / 60, 1 / 30, 1 / 144]))))]
    kind = int(r.integers(0, 5))
    if kind == 0:    # digital silence until the display decays and the source goes silent, then noise again
        steps += [("silence", n + 400), ("tick",)] + [("silence", 800), ("tick",)] * 12 + [("noise", 800), ("tick",)] * 2
    elif kind == 1 and cfg["capture_channels"] == 2:
        steps += [("noise_ch0_only", n + 400), ("tick",)] + [("noise_ch0_only", 800), ("tick",)] * 8
    elif kind == 2:
        steps += [("hide",), ("noise", 800), ("tick",), ("tick",), ("show",), ("noise", 800), ("tick",), ("noise", 800), ("tick",)]
    elif kind == 3:
        steps += [("mute", 800), ("tick",), ("noise", 800), ("tick",), ("mute", n), ("tick",)]
    # a quarter of the cases run with an audio sync offset: a constant A/V-sync reserve of sync_ms * 48 frames behind the
    # window (dtaudio > 0, src/source_generic.cpp:50-59; sync_rms_buffer holds the RMS values back as well)
    sync_ms = int(r.choice([0, 0, 0, 5, 20]))
    # (drawn last so that the cases of earlier rounds keep everything else) a third of the display cases also fill the vertex
    # buffer: bars as triangles, the curve as a triangle strip or a line strip
    if display and r.random() < 0.35:
        cfg.update(vertices=int(r.choice([1, 3])) if display == 1 else int(r.integers(1, 3)))
        if cfg["vertices"] == 3:
            cfg.update(step_width=int(r.choice([8, 3, 12])), step_gap=int(r.choice([4, 1, 0])), rounded_caps=0)
        elif display == 1 and cfg.get("rounded_caps") and r.random() < 0.5:
            cfg.update(radial=1)  # full-circle cap fans
    # (drawn after everything else, end of round 2) OBS's other audio rate, and display ranges other than 30 Hz - 17.5 kHz
    if r.random() < 0.25:
        cfg.update(sample_rate=44100)
    if display and r.random() < 0.3:
        cfg.update(cutoff_low=int(r.choice([20, 60, 120, 500])), cutoff_high=int(r.choice([6000, 12000, 20000, 22000])))
    return cfg, steps, sync_ms


WIDE_SEEDS = range(400)    # the reference's full slider ranges (draw_wide)


def draw_wide(seed: int):
    """Every dimension over the range the reference's own property sliders allow (/root/reference/src/source.cpp:195-447):
    gravity 0 ... 1.0 inclusive (get_gravity's special case at 0, src/source.hpp:301-312), filter radius 0 ... 32, width
    32 ... 3840, height 32 ... 2160, bar width 1 ... 256, bar gap 0 ... 256, step width / gap likewise, minimum bar height
    0 ... 1080, channel spacing 0 ... 2160, sine exponent 1 ... 16, floor and ceiling anywhere in -120 ... 0 (get_settings
    repairs ceiling <= floor), slope 0 ... 10, roll-off Q 0 ... 10 and rate 0 ... 65, cut-offs 0 ... 24000 Hz in any order,
    volume target -60 ... 0, maximum gain 0 ... 45.  Always with a display (that is where the wide ranges bite)."""
    r = np.random.default_rng(31337000 + seed)

    def edge(lo, hi, integer=True, p_edge=0.3):
        """a value in [lo, hi], the two ends themselves a good part of the time"""
        u = r.random()
        if u < p_edge / 2:
            return lo
        if u < p_edge:
            return hi
        return int(r.integers(lo, hi + 1)) if integer else float(np.float32(round(float(r.uniform(lo, hi)), 2)))

    if r.random() < 0.75:
        n = int(r.choice([128, 256, 512, 1024, 2048, 4096, 8192, 16384], p=[0.1, 0.1, 0.1, 0.25, 0.2, 0.15, 0.06, 0.04]))
    else:
        n = 16 * int(r.integers(8, 4096 // 16 + 1))
    layout = int(r.integers(0, 4))
    cfg = dict(fft_size=n, capture_channels=1 if layout in (0, 3) else 2, stereo=1 if layout in (2, 3) else 0,
               window=int(r.integers(0, 6)), sine_exponent=edge(1, 16), tsmoothing=int(r.integers(0, 3)),
               gravity=edge(0.0, 1.0, integer=False, p_edge=0.4), fast_peaks=int(r.integers(0, 2)),
               slope=0.0 if r.random() < 0.4 else edge(0.0, 10.0, integer=False),
               floor_db=edge(-120, 0, p_edge=0.15), ceiling_db=edge(-120, 0, p_edge=0.15))
    if r.random() < 0.35:
        cfg.update(rolloff_q=edge(0.0, 10.0, integer=False), rolloff_rate=edge(0.0, 65.0, integer=False))
    if r.random() < 0.3:
        cfg.update(normalize_volume=1, volume_target=float(edge(-60, 0)), max_gain=float(edge(0, 45)))
    display = int(r.integers(1, 3))
    cfg.update(interp_mode=int(r.integers(0, 3)), log_scale=int(r.integers(0, 2)), mirror_freq_axis=int(r.random() < 0.3),
               width=edge(32, 3840) if r.random() < 0.6 else int(r.choice([800, 1920, 2560])),
               height=edge(32, 2160) if r.random() < 0.5 else int(r.choice([225, 300, 1080])),
               channel_spacing=int(r.choice([0, 0, 1, 6])) if r.random() < 0.8 else edge(0, 2160))
    if display == 1:
        cfg.update(bars=1, bar_width=edge(1, 256) if r.random() < 0.5 else int(r.choice([1, 2, 3, 5, 24])),
                   bar_gap=edge(0, 256) if r.random() < 0.4 else int(r.choice([0, 1, 6])),
                   min_bar_height=int(r.choice([0, 3])) if r.random() < 0.8 else edge(0, 1080), rounded_caps=int(r.random() < 0.3))
    else:
        cfg.update(curve=1)
    if r.random() < 0.6:
        cfg.update(filter_mode=1, filter_radius=edge(0.0, 32.0, integer=False))
    if r.random() < 0.3:
        cfg.update(cutoff_low=edge(0, 24000, p_edge=0.2), cutoff_high=edge(0, 24000, p_edge=0.2))
    if r.random() < 0.3:
        cfg.update(vertices=int(r.choice([1, 3])) if display == 1 else int(r.integers(1, 3)))
        if cfg["vertices"] == 3:
            # (channel spacing stays below the channel's height here: create_vbuf converts (cpos - channel_offset) / step_stride
            # to size_t, undefined for a negative quotient, src/source.cpp:996)
            cfg.update(step_width=edge(1, 256) if r.random() < 0.4 else int(r.choice([8, 3, 1])),
                       step_gap=edge(0, 256) if r.random() < 0.4 else int(r.choice([4, 1, 0])), rounded_caps=0,
                       channel_spacing=int(r.choice([0, 1, 6])))
        elif display == 1 and cfg.get("rounded_caps") and r.random() < 0.5:
            cfg.update(radial=1)
    if r.random() < 0.25:
        cfg.update(sample_rate=44100)
    steps = []
    for _ in range(int(r.integers(3, 6))):
        steps += [("noise", int(r.choice([800, 441, 1024, 37, 1600]))), ("tick", float(np.float32(r.choice([1 / 60, 1 / 30, 1 / 144]))))]
    kind = int(r.integers(0, 4))
    if kind == 0:
        steps += [("silence", n + 400), ("tick",)] + [("silence", 800), ("tick",)] * 10 + [("noise", 800), ("tick",)] * 2
    elif kind == 1:
        steps += [("hide",), ("noise", 800), ("tick",), ("show",), ("noise", 800), ("tick",)]
    sync_ms = int(r.choice([0, 0, 0, 20]))
    return cfg, steps, sync_ms


def _undo_db(cfg):
    """per-bin dB offsets applied after the FFT (the roll-off table), added back before the linear-domain comparison"""
    if not (cfg.rolloff_q > 0 and cfg.rolloff_rate > 0):
        return None
    from oracle import restate
    o = restate.OracleSource(cfg)
    try:
        ro = o.rolloff()
    finally:
        o.close()
    if ro is None:
        return None
    ro = ro.astype(np.float64)
    ro[0] = 0.0  # the roll-off loop starts at bin 1 (src/source_generic.cpp:173)
    return ro


def _render_from_rows(cfg, rows):
    """what the restated render_bars / render_curve makes of the given dB rows: (bars [ch, n], vertices per channel or None)"""
    import ctypes as C
    from oracle import restate
    o = restate.OracleSource(cfg)
    try:
        rows = np.ascontiguousarray(rows, np.float32)
        for c in range(o.display_channels):
            C.memmove(o.L.wfo_decibels(o.h, c), rows[c].ctypes.data, rows[c].nbytes)
        o.render_bars()
        bars = o.bars().copy()
        verts = [o.vertices(c, line=cfg.vertices == 2).copy() for c in range(o.display_channels)] if cfg.vertices else None
    finally:
        o.close()
    return bars, verts


# How often the second arm of the display check is taken (cases whose bars / curve miss the reference's by more than the pixel
# tolerance and are then held against the render of the device's own rows): counted, reported, and bounded by
# test_zz_display_arm_stays_rare at the end of this module -- a regression that leans on the arm shows up as a count.
ARM = {"display_checks": 0, "display_arm": 0, "arm_cases": [], "unsupported": []}


def _px_tol(cfg):
    """absolute pixel tolerance of a display value: 2e-3 px, which at up to 20 px per dB is the rows' own absolute tolerance
    (1e-4 dB, helpers.ATOL); a display that stretches its dB range over more pixels than that -- the reference's sliders allow
    2160 px over 1 dB -- is held to the same 1e-4 dB, i.e. proportionally more pixels"""
    if cfg is None:
        return 2e-3
    rng = float(cfg.ceiling_db - cfg.floor_db)
    if rng <= 0:
        rng = 120.0  # get_settings repairs ceiling <= floor to 0 / -120 (src/source.cpp:572-576)
    return 2e-3 * max(1.0, (float(cfg.height) / rng) / 20.0)


def _quads_by_bar(v):
    """stepped bars: the step quads (6 vertices each) grouped by the x of their bar, bottom step first"""
    out = {}
    for q in range(v.shape[0] // 6):
        quad = v[q * 6:(q + 1) * 6]
        out.setdefault(float(quad[:, 0].min()), []).append(quad)
    return out


def _compare(got, want, undo, what, cfg_stepped=False, cfg=None):
    """rows against rows (assert_db_close), then the display derived from them.  A bar or curve point averages dB values, and a
    bin in a deep null may differ by whole dB between two correct float FFTs (the linear arm of assert_db_close allows it): where
    the display misses the reference's by more than the pixel tolerance it is held, with the same tolerance, against what the
    restated render loop makes of the *device's own rows* -- a stage is not faulted for the latitude of the stage before it."""
    assert len(got) == len(want)
    px = _px_tol(cfg)
    for t, (g, w) in enumerate(zip(got, want)):
        assert g["silent"] == w["silent"], f"{what} tick {t}: m_last_silent {g['silent']} != {w['silent']}"
        assert_db_close(g["db"], w["db"], f"{what} tick {t} decibels", undo_db=undo, deep=True)
        ref_bars, ref_verts = w["bars"], w.get("verts")
        assert (g["bars"] is None) == (w["bars"] is None), f"{what} tick {t}: one side has no bars"
        if w["bars"] is not None:
            err = np.abs(g["bars"].astype(np.float64) - w["bars"])
            ARM["display_checks"] += 1
            if not np.all(err <= 1e-5 * np.abs(w["bars"]) + px) and cfg is not None:
                ARM["display_arm"] += 1
                ARM["arm_cases"].append(f"{what} tick {t}: {err.max():.2e} px")
                ref_bars, alt_verts = _render_from_rows(cfg, g["db"])
                ref_verts = alt_verts if alt_verts is not None else ref_verts
                err = np.abs(g["bars"].astype(np.float64) - ref_bars)
            assert np.all(err <= 1e-5 * np.abs(ref_bars) + px), f"{what} tick {t} bars/curve: max err {err.max():.3e} px (tolerance {px:.1e})"
        if "verts" in w:
            for c, (gv, wv) in enumerate(zip(g["verts"], ref_verts)):
                # stepped bars: a bar whose height sits within rounding of a step boundary may gain or lose that step
                if gv.shape != wv.shape and cfg_stepped:
                    # bar by bar: the same bars, at most one step apart, and the steps both sides have are the same quads
                    gq, wq = _quads_by_bar(gv), _quads_by_bar(wv)
                    assert abs(gv.shape[0] - wv.shape[0]) <= 6 * 2, f"{what} tick {t} channel {c}: {gv.shape[0]} vs {wv.shape[0]} vertices"
                    for x in sorted(set(gq) | set(wq)):
                        a, b = gq.get(x, []), wq.get(x, [])
                        assert abs(len(a) - len(b)) <= 1, f"{what} tick {t} channel {c}: bar at x={x} has {len(a)} vs {len(b)} steps"
                        for qa, qb in zip(a, b):
                            assert np.array_equal(qa[:, 0], qb[:, 0]) and np.all(np.abs(qa[:, 1].astype(np.float64) - qb[:, 1]) <= 1e-5 * np.abs(qb[:, 1]) + px), \
                                f"{what} tick {t} channel {c}: a step of the bar at x={x} differs"
                    continue
                assert gv.shape == wv.shape, f"{what} tick {t} channel {c}: vertex count {gv.shape} vs {wv.shape}"
                assert np.array_equal(gv[..., 0], wv[..., 0]), f"{what} tick {t} channel {c}: vertex x"
                err = np.abs(gv[..., 1].astype(np.float64) - wv[..., 1])
                assert np.all(err <= 1e-5 * np.abs(wv[..., 1]) + px), f"{what} tick {t} channel {c} vertex y: max err {err.max():.3e} px"
        if "rms" in w:
            assert abs(float(g["rms"]) - float(w["rms"])) <= 1e-5 * abs(float(w["rms"])) + 1e-9, f"{what} tick {t} m_input_rms"


def run_spectrum_case(seed, family, fft_size=None):
    import waveform_amd as wf
    from oracle import wfref
    cfg_dict, steps, sync_ms = draw_wide(seed) if family == "wide" else draw(seed, family, fft_size)
    cfg = scenarios.make_config(cfg_dict)
    sc = dict(cfg=cfg_dict, steps=steps, record="all", sync_ms=sync_ms)
    what = f"{family} case {seed} ({cfg_dict}, sync {sync_ms} ms)"
    undo = _undo_db(cfg)
    stepped = cfg_dict.get("vertices") == 3
    rms = 0.0316 if cfg.normalize_volume else 0.0  # what the host's update_input_rms would hand over (-30 dBFS)
    # the restatement's double DFT is O(n p) for a length with largest prime factor p: beyond RESTATEMENT_MAX_PRIME the
    # reference itself (FFTW) is the only checker -- which needs the reference library
    restatement = _largest_prime_factor(int(cfg.fft_size)) <= RESTATEMENT_MAX_PRIME
    if not restatement:
        assert wfref.available(), "oracle/_ref/libwfref.so is needed to check this length (largest prime factor too large for the restatement)"
    if restatement:
        try:
            hip = scenarios.HipBackend(cfg, streams=2, probe=1, input_rms=rms)
        except wf.WfHipError as e:
            # every drawn configuration is legal for the reference: the library either takes it or says what it does not take
            # (WF_HIP_ERR_UNSUPPORTED with a text); anything else is a failure.  Recorded; bounded at the end of the module.
            assert e.code == -2 and len(str(e)) > 30, f"{what}: {e}"
            ARM["unsupported"].append(f"{what}: {e}")
            return
        ora = scenarios.OracleBackend(cfg, input_rms=rms)
        try:
            got = scenarios.play(hip, sc)
            want = scenarios.play(ora, sc)
        finally:
            hip.close()
        _compare(got, want, undo, what + " vs the restatement", cfg_stepped=stepped, cfg=cfg)
    if (seed % REF_EVERY == 0 or not restatement) and wfref.available():
        # the same script against the reference itself (its own float FFTW, its own update_input_rms); the device derives
        # m_input_rms from the audio too
        try:
            hip = scenarios.HipBackend(cfg, streams=2, probe=0)
        except wf.WfHipError as e:
            assert e.code == -2 and len(str(e)) > 30, f"{what}: {e}"
            ARM["unsupported"].append(f"{what}: {e}")
            return
        ref = scenarios.RefBackend(cfg)
        try:
            got = scenarios.play(hip, sc)
            want = scenarios.play(ref, sc)
        finally:
            hip.close()
        _compare(got, want, undo, what + " vs libwfref", cfg_stepped=stepped, cfg=cfg)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", SPEC_SEEDS)
def test_hip_matches_oracle_on_random_case(seed):
    run_spectrum_case(seed, "pow2")


@pytest.mark.gpu
@pytest.mark.parametrize("seed", HUGE_SEEDS)
def test_hip_matches_oracle_on_random_huge_size(seed):
    run_spectrum_case(seed, "huge")


@pytest.mark.gpu
@pytest.mark.parametrize("seed", BLU_SEEDS)
def test_hip_matches_oracle_on_random_size(seed):
    run_spectrum_case(seed, "any")


@pytest.mark.gpu
@pytest.mark.parametrize("seed", SMOOTH_SEEDS)
def test_hip_matches_oracle_on_random_smooth_size(seed):
    run_spectrum_case(seed, "smooth")


@pytest.mark.gpu
@pytest.mark.parametrize("n", [800, 960, 720, 880, 1600, 1920, 2000, 1760])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_automatic_sizes_with_compile_time_plans(n, seed):
    """The sizes the plugin picks by itself (sample_rate / fps & -16; reference src/source.cpp:1161-1166) run their mixed-radix plan as
    compile-time constants (mr_transform_fixed, wf_mixed.hpp): each of them through drawn scenarios of the smooth family"""
    run_spectrum_case(7000 + 10 * seed + n, "smooth", fft_size=n)


@pytest.mark.gpu
def test_smooth_sizes_take_the_mixed_radix_kernel_and_the_others_bluestein():
    import waveform_amd as wf
    for n, mixed in ((800, True), (1600, True), (960, True), (8000, True), (16320, True), (16336, False), (4160, True), (1760, True), (1456, True), (1824, True), (1088, True), (1472, True), (464, True), (4144, True), (7808, True), (8128, False), (2096, False), (13456, False), (144, True), (15552, True)):
        with wf.SpectrumBatch(wf.Config.defaults(fft_size=n), 2) as b:
            name = b.kernel_name()
            assert ("mixed radix" in name) == mixed and ("Bluestein" in name) != mixed, (n, name)
    # above 16384: rows of a mixed-radix transform where n/2 = C R has a plan without a prime pass (two rows: both and the end of the
    # tick in one kernel), else (n a multiple of 16 -- every
    # position of the reference's slider is one of 64) rows by Bluestein inside LDS (16400 = 2 x 41 x 10 x 10 included).  Every legal size
    # takes one of the two: Bluestein through device memory is left to the development builds' WF_HIP_NO_BLUESTEIN_ROWS=1
    for n, kernel in ((48000, "big_mr_rows_kernel"), (32000, "big_mr_whole_kernel"), (65520, "big_mr_rows_kernel"), (20480, "big_mr_whole_kernel"), (30000, "big_mr_whole_kernel"),
                      (16400, "big_br_{columns,rows}_kernel"), (48016, "big_br_{columns,rows}_kernel"), (33824, "big_br_{columns,rows}_kernel"),
                      (65424, "big_br_{columns,rows}_kernel"), (17488, "big_br_{columns,rows}_kernel")):
        with wf.SpectrumBatch(wf.Config.defaults(fft_size=n), 1) as b:
            name = b.kernel_name()
            assert name.startswith(kernel + " ") or name.startswith(kernel + "<"), (n, name)
            assert ("Bluestein" in name) == (not kernel.startswith("big_mr_")), (n, name)
    # 16 rows where n/2 is a multiple of 16 (every slider position), 8 for the other multiples of 16; the container: >= 2 R - 1 points
    for n, rows, r, container in ((48064, 16, 1502, 4096), (65472, 16, 2046, 4096), (32704, 16, 1022, 2048), (17488, 8, 1093, 4096), (48016, 8, 3001, 8192)):
        with wf.SpectrumBatch(wf.Config.defaults(fft_size=n), 1) as b:
            assert f"{rows} rows of {r} complex points by Bluestein over {container} points" in b.kernel_name(), (n, b.kernel_name())


@pytest.mark.gpu
@pytest.mark.parametrize("seed", WIDE_SEEDS)
def test_hip_matches_oracle_over_the_full_slider_ranges(seed):
    run_spectrum_case(seed, "wide")


@pytest.mark.gpu
def test_per_stream_input_rms():
    """wf_hip_set_input_rms: every stream of a batch is normalised with its own m_input_rms
    (volume_compensation, reference src/source_generic.cpp:161-167)"""
    import waveform_amd as wf
    from tools import synth
    cfg_dict = dict(fft_size=2048, stereo=1, normalize_volume=1, volume_target=-8.0, max_gain=30.0, slope=1.0)
    cfg = scenarios.make_config(cfg_dict)
    rms = np.array([0.5, 0.0316, 0.0, 1e-4], np.float32)  # 0.0: "no audio seen yet" -> the full max_gain
    audio = synth.block(scenarios.SEED, 0, 1, 2, 0, 800 * 4)[0]
    with wf.SpectrumBatch(cfg, len(rms)) as b:
        b.set_input_rms(rms[:2])
        b.set_input_rms(rms[2:], first=2)
        for t in range(4):
            b.push_audio(np.broadcast_to(audio[None, :, t * 800:(t + 1) * 800], (len(rms), 2, 800)))
            b.tick(input_rms=0.123)  # ignored once per-stream values exist
        got = b.decibels()
    for i, r in enumerate(rms):
        ora = scenarios.OracleBackend(cfg, input_rms=float(r))
        for t in range(4):
            ora.push(audio[:, t * 800:(t + 1) * 800], muted=False)
            ora.tick(1.0 / 60.0)
        assert_db_close(got[i], ora.observe()["db"], f"stream {i} (input_rms {r})", deep=True)
    assert not np.allclose(got[0], got[1])


@pytest.mark.gpu
def test_per_stream_av_sync_delay():
    """wf_hip_set_stream_delay: every stream analyses the window ending its own number of frames before its newest sample
    (dtaudio > 0 of each source, reference src/source_generic.cpp:50-59), aligned and unaligned delays"""
    import waveform_amd as wf
    from tools import synth
    cfg_dict = dict(fft_size=1024, stereo=1, tsmoothing=0)
    cfg = scenarios.make_config(cfg_dict)
    for delays in ([0, 400, 800, 1200], [0, 37, 441, 1023]):
        d = np.array(delays, np.uint32)
        total = 1024 + 1600
        audio = synth.block(scenarios.SEED, 0, 1, 2, 0, total)[0]
        with wf.SpectrumBatch(cfg, len(d), ring_frames=4096) as b:
            b.set_stream_delay(d)
            b.push_audio(np.broadcast_to(audio[None], (len(d), 2, total)))
            b.tick(delay_frames=100 if delays[1] == 37 else 0)
            got = b.decibels()
        common = 100 if delays[1] == 37 else 0
        for i, di in enumerate(delays):
            ora = scenarios.OracleBackend(cfg)
            ora.push(audio[:, : total - di - common], muted=False)  # the oracle sees the audio up to the window's end
            ora.tick(1.0 / 60.0)
            assert_db_close(got[i], ora.observe()["db"], f"delays {delays}: stream {i}", deep=True)


# ---- level meter --------------------------------------------------------------------------------------------------------
METER_SEEDS = range(500)


def draw_meter(seed: int):
    r = np.random.default_rng(5000 + seed)
    cfg = dict(meter=1, meter_rms=int(r.integers(0, 2)), meter_ms=int(r.choice([20, 50, 100, 150, 333, 500])),
               capture_channels=int(r.integers(1, 3)), tsmoothing=int(r.integers(0, 3)),
               gravity=float(np.float32(r.uniform(0.05, 0.95))), fast_peaks=int(r.integers(0, 2)),
               floor_db=int(r.choice([-65, -80, -50])), ceiling_db=int(r.choice([0, -6])), height=int(r.choice([225, 300, 101])),
               rounded_caps=int(r.random() < 0.3), min_bar_height=int(r.choice([0, 3])), bar_width=int(r.choice([24, 12])))
    steps = []
    for _ in range(int(r.integers(18, 30))):  # long enough for the EMA to climb out of its DB_MIN start
        steps += [("noise_amp", int(r.choice([800, 441, 1024, 37, 1600])), float(np.float32(r.choice([1.0, 0.3, 0.01])))),
                  ("tick", float(np.float32(r.choice([1 / 60, 1 / 30, 1 / 144]))))]
    kind = int(r.integers(0, 4))
    if kind == 0:
        steps += [("silence", 800), ("tick",)] * 14 + [("noise", 800), ("tick",)] * 3
    elif kind == 1:
        steps += [("hide",), ("noise", 800), ("tick",), ("tick",), ("show",), ("noise", 800), ("tick",), ("noise", 800), ("tick",)]
    elif kind == 2:
        steps += [("timeout",), ("tick",), ("tick",), ("noise", 800), ("tick",), ("noise", 800), ("tick",)]
    elif cfg["capture_channels"] == 2:
        steps += [("noise_ch0_only", 800), ("tick",)] * 10
    if r.random() < 0.25:  # (drawn last) OBS's other audio rate: the meter buffer is sample_rate * meter_ms long
        cfg.update(sample_rate=44100)
    return cfg, steps


METER_WIDE_SEEDS = range(60)


def draw_meter_wide(seed: int):
    """the level meter over the reference's slider ranges: meter_buf 10 ms ... 60 s (the property allows 600 s,
    src/source.cpp:323: test_meter_buffer_up_to_the_reference_maximum plays that once), gravity 0 ... 1.0 inclusive, floor and
    ceiling anywhere, bar geometry anywhere"""
    r = np.random.default_rng(77000 + seed)
    cfg = dict(meter=1, meter_rms=int(r.integers(0, 2)), meter_ms=int(r.choice([10, 20, 1000, 5000, 20000, 60000])),
               capture_channels=int(r.integers(1, 3)), tsmoothing=int(r.integers(0, 3)),
               gravity=float(np.float32(r.choice([0.0, 1.0, round(float(r.uniform(0, 1)), 2)]))), fast_peaks=int(r.integers(0, 2)),
               floor_db=int(r.integers(-120, 1)), ceiling_db=int(r.integers(-120, 1)), height=int(r.choice([32, 225, 2160])),
               rounded_caps=int(r.random() < 0.3), min_bar_height=int(r.choice([0, 3, 400])), bar_width=int(r.choice([1, 24, 256])))
    steps = []
    for _ in range(int(r.integers(8, 14))):
        steps += [("noise_amp", int(r.choice([800, 441, 1024, 4800, 48000])), float(np.float32(r.choice([1.0, 0.3, 0.01])))),
                  ("tick", float(np.float32(r.choice([1 / 60, 1 / 30, 1 / 144]))))]
    kind = int(r.integers(0, 3))
    if kind == 0:
        steps += [("hide",), ("noise", 800), ("tick",), ("show",), ("noise", 800), ("tick",)]
    elif kind == 1:
        steps += [("timeout",), ("tick",), ("noise", 800), ("tick",)]
    if r.random() < 0.25:
        cfg.update(sample_rate=44100)
    return cfg, steps


def run_meter_case(seed, wide=False):
    cfg_dict, steps = draw_meter_wide(seed) if wide else draw_meter(seed)
    cfg = scenarios.make_config(cfg_dict)
    sc = dict(cfg=cfg_dict, steps=steps, record="all")
    hip = scenarios.HipBackend(cfg, streams=3, probe=2)
    ora = scenarios.OracleBackend(cfg)
    exact = scenarios.OracleBackend(cfg, exact=True)  # the same restatement with the sum of squares in double
    try:
        got = scenarios.play(hip, sc)
        want = scenarios.play(ora, sc)
        truth = scenarios.play(exact, sc)
    finally:
        hip.close()
    assert len(got) == len(want) == len(truth)
    dbrange = float(cfg.ceiling_db - cfg.floor_db)
    for t, (g, w, x) in enumerate(zip(got, want, truth)):
        what = f"meter case {seed} tick {t} ({cfg_dict})"
        # m_last_silent is a threshold on the level: it may only differ where the reference and the exact sum differ too
        assert g["silent"] == w["silent"] or g["silent"] == x["silent"], f"{what}: m_last_silent {g['silent']} != {w['silent']}"
        assert_levels_close(g["db"], w["db"], x["db"], what + " levels")
        tol = 1e-5 * np.abs(w["bars"]) + 2e-3
        gb, wb, xb = (np.asarray(v["bars"], np.float64) for v in (g, w, x))
        ok = (np.abs(gb - wb) <= tol) | (np.abs(gb - xb) <= np.abs(wb - xb) + tol)
        assert ok.all(), f"{what} bars: got {gb}, reference {wb}, exact sum {xb} (range {dbrange} dB)"


@pytest.mark.gpu
@pytest.mark.parametrize("seed", METER_SEEDS)
def test_hip_meter_matches_oracle_on_random_case(seed):
    run_meter_case(seed)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", METER_WIDE_SEEDS)
def test_hip_meter_matches_oracle_over_the_full_slider_ranges(seed):
    run_meter_case(seed, wide=True)


@pytest.mark.gpu
def test_meter_buffer_up_to_the_reference_maximum():
    """meter_buf = 600 000 ms, the property's maximum (src/source.cpp:323): a meter buffer of 28.8 M samples per channel -- a
    512 MB ring per stereo stream on the device.  Levels after each of a few one-second packets against the restatement
    (and its exactly summed twin: the reference's sequential float sum over 28.8 M squares drifts by itself)."""
    import waveform_amd as wf
    from oracle import restate
    from tools import synth
    cfg = wf.Config.defaults(meter=1, meter_rms=1, meter_ms=600000, tsmoothing=0)
    ora, exact = restate.OracleMeter(cfg), restate.OracleMeter(cfg)
    exact.set_exact(True)
    with wf.SpectrumBatch(cfg, 2) as b:
        assert b.fft_size == (48000 * 600) & ~15
        for t in range(3):
            audio = synth.block(synth.DEFAULT_SEED, 0, 1, 2, t * 48000, 48000) * np.float32(0.5 if t == 1 else 1.0)
            b.push_audio(np.broadcast_to(audio, (2, 2, 48000)))
            b.tick()
            for o in (ora, exact):
                o.push_audio(audio[0])
                o.tick()
            got = b.meter()
            assert np.array_equal(got[0], got[1])
            assert_levels_close(got[0], ora.levels(), exact.levels(), f"600 s meter buffer, packet {t}")


@pytest.mark.gpu
def test_meter_sync_delay_never_unconsumes():
    """tick_meter pops everything older than the A/V-sync point; a later tick with a larger delay leaves the meter buffer
    as it was (src/source_generic.cpp:204-220: the while loop simply does not run)."""
    import waveform_amd as wf
    from oracle import restate
    from tools import synth
    cfg = wf.Config.defaults(meter=1, meter_ms=50, tsmoothing=0)
    ora = restate.OracleMeter(cfg)
    delays = [0, 400, 1300, 100, 0, 2000, 5, 0]
    with wf.SpectrumBatch(cfg, 2, ring_frames=16384) as b:
        for t, d in enumerate(delays):
            audio = synth.block(synth.DEFAULT_SEED, 0, 1, 2, t * 800, 800)
            b.push_audio(np.broadcast_to(audio, (2, 2, 800)))
            ora.set_sync_delay(d)
            ora.push_audio(audio[0])
            b.tick(delay_frames=d)
            ora.tick()
            assert_db_close(b.meter()[1], ora.levels(), f"tick {t} delay {d}")


# ---- waveform display -----------------------------------------------------------------------------------------------------
WAVE_SEEDS = range(100)


def draw_wave(seed: int):
    r = np.random.default_rng(9000 + seed)
    layout = int(r.integers(0, 4))
    cfg = dict(waveform=1, capture_channels=1 if layout in (0, 3) else 2, stereo=1 if layout in (2, 3) else 0,
               width=int(r.choice([800, 333, 1024, 64, 2000])), meter_ms=int(r.choice([150, 50, 400, 20])))
    if r.random() < 0.4:
        cfg.update(normalize_volume=1, volume_target=float(r.integers(-20, -2)), max_gain=float(r.integers(6, 31)))  # integer sliders in the reference
    steps = []
    for _ in range(int(r.integers(6, 14))):
        for _ in range(int(r.integers(0, 3))):
            steps.append(("noise_amp", int(r.choice([800, 441, 1024, 37])), float(np.float32(r.choice([1.0, 0.2])))))
        steps.append(("tick",))
    kind = int(r.integers(0, 3))
    if kind == 0:
        steps += [("hide",), ("noise", 800), ("tick",), ("tick",), ("show",), ("noise", 800), ("tick",)]
    elif kind == 1:
        steps += [("timeout",), ("tick",), ("noise", 800), ("tick",), ("noise", 800), ("tick",)]
    else:
        steps += [("silence", 1024)] * 8 + [("tick",), ("noise", 800), ("tick",)]
    sync = int(r.choice([0, 0, 5, 20]))
    if r.random() < 0.25:  # (drawn last) OBS's other audio rate
        cfg.update(sample_rate=44100)
    return cfg, steps, sync


@pytest.mark.gpu
@pytest.mark.parametrize("seed", WAVE_SEEDS)
def test_hip_waveform_matches_oracle_on_random_case(seed):
    cfg_dict, steps, sync_ms = draw_wave(seed)
    cfg = scenarios.make_config(cfg_dict)
    sc = dict(cfg=cfg_dict, steps=steps, record="all", sync_ms=sync_ms)
    rms = 0.0316 if cfg.normalize_volume else 0.0
    hip = scenarios.HipBackend(cfg, streams=3, probe=0, input_rms=rms)
    ora = scenarios.OracleBackend(cfg, input_rms=rms)
    try:
        got = scenarios.play(hip, sc)
        want = scenarios.play(ora, sc)
    finally:
        hip.close()
    assert len(got) == len(want)
    for t, (g, w) in enumerate(zip(got, want)):
        assert g["silent"] == w["silent"], f"wave case {seed} tick {t}: m_last_silent {g['silent']} != {w['silent']} ({cfg_dict})"
        assert_db_close(g["db"], w["db"], f"wave case {seed} tick {t} rows ({cfg_dict}, sync {sync_ms} ms)", lin_eps=None)
        assert g["wts"] == w["wts"], f"wave case {seed} tick {t}: m_waveform_ts {g['wts']} != {w['wts']} ({cfg_dict}, sync {sync_ms} ms)"


# ---- the same scripts through the reference plugin with WAVSourceHIP plugged in (the drop-in binding) -------------------------
def run_dropin_case(seed, family):
    """the fuzz script of `family` through the reference plugin itself, once with its own CPU class and once with WAVSourceHIP
    (synchronous mode) as the tick implementation; the device path must stay in use, no tick may fall back"""
    import numpy as np
    from pathlib import Path
    import scenarios
    from oracle import wfref
    from helpers import assert_db_close
    import os
    os.environ["WF_HIP_LIBRARY"] = str(Path(__file__).resolve().parent.parent / "waveform_amd" / "libwaveform_hip.so")
    os.environ["WF_HIP_BATCHED"] = "0"
    if family == "meter":
        cfg_dict, steps = draw_meter(seed)
        sync_ms = 0
    elif family == "wave":
        cfg_dict, steps, sync_ms = draw_wave(seed)
    else:
        cfg_dict, steps, sync_ms = draw(seed, family)
    cfg_dict = dict(cfg_dict)   # (vertices stay: WAVSourceHIP::render draws the plugin's vertex buffer from the device's)
    cfg = scenarios.make_config(cfg_dict)
    sc = dict(cfg=cfg_dict, steps=steps, record="all", sync_ms=sync_ms)
    before = wfref.hip_fallback_ticks()
    drawn, on_host = wfref.hip_device_renders(), wfref.hip_host_renders()
    hip = scenarios.RefBackend(cfg, isa="hip")
    assert hip.src.using_hip, "WAVSourceHIP did not take the device path"
    got = scenarios.play(hip, sc)
    assert hip.src.using_hip and wfref.hip_fallback_ticks() == before, "fell back to the CPU class"
    if family not in ("meter", "wave"):
        import test_golden as tg
        tg._check_renders(wfref, cfg, drawn, on_host)
    want = scenarios.play(scenarios.RefBackend(cfg, isa="generic"), sc)
    assert len(got) == len(want)
    undo = _undo_db(cfg) if family in ("pow2", "any", "huge") else None
    truth = scenarios.play(scenarios.OracleBackend(cfg, exact=True), sc) if family == "meter" else [None] * len(want)
    for t, (g, w, x) in enumerate(zip(got, want, truth)):
        what = f"drop-in {family} case {seed} tick {t} ({cfg_dict}, sync {sync_ms} ms)"
        if family == "meter":
            # the criterion of the batch fuzz (helpers.assert_levels_close): within tolerance of the reference, or no farther
            # from the exactly summed level than the reference's own sequential float sum is
            assert g["silent"] == w["silent"] or g["silent"] == x["silent"], what + f": m_last_silent {g['silent']} != {w['silent']}"
            from helpers import assert_levels_close
            assert_levels_close(g["db"], w["db"], x["db"], what + " levels")
            continue
        assert g["silent"] == w["silent"], what + f": m_last_silent {g['silent']} != {w['silent']}"
        if family in ("pow2", "any", "huge"):
            # rows, and what render() drew from the device's display: bar tops / curve points and the vertex buffer of every gs_draw
            _compare([g], [w], undo, what, cfg_stepped=cfg_dict.get("vertices") == 3, cfg=cfg)
        else:
            assert_db_close(g["db"], w["db"], what + " rows", undo_db=undo, lin_eps=None)


def run_dropin_batched_case(seed, family):
    """spectrum scripts through the plugin's batched mode (sources share a handle, rows read one frame late)"""
    import numpy as np
    from pathlib import Path
    import scenarios
    import test_golden as tg  # (_OneFrameLate)
    from oracle import wfref
    from helpers import assert_db_close
    import os
    os.environ["WF_HIP_LIBRARY"] = str(Path(__file__).resolve().parent.parent / "waveform_amd" / "libwaveform_hip.so")
    os.environ["WF_HIP_BATCHED"] = "1"
    cfg_dict, steps, sync_ms = draw(seed, family)
    cfg_dict = dict(cfg_dict)
    cfg = scenarios.make_config(cfg_dict)
    sc = dict(cfg=cfg_dict, steps=steps, record="all", sync_ms=sync_ms)
    before = wfref.hip_fallback_ticks()
    drawn, on_host = wfref.hip_device_renders(), wfref.hip_host_renders()
    late = tg._OneFrameLate(scenarios.RefBackend(cfg, isa="hip"))
    assert late.be.src.using_hip
    scenarios.play(late, sc)
    got = late.finish()
    assert late.be.src.using_hip and wfref.hip_fallback_ticks() == before, "fell back to the CPU class"
    tg._check_renders(wfref, cfg, drawn, on_host)
    want = scenarios.play(scenarios.RefBackend(cfg, isa="generic"), sc)
    assert len(got) == len(want), (len(got), len(want))
    undo = _undo_db(cfg)
    for t, (g, w) in enumerate(zip(got, want)):
        what = f"batched drop-in {family} case {seed} tick {t} ({cfg_dict}, sync {sync_ms} ms)"
        _compare([g], [w], undo, what, cfg_stepped=cfg_dict.get("vertices") == 3, cfg=cfg)  # rows, bars / curve, vertex buffers: one frame late




def run_dropin_batched_meter_case(seed):
    """level-meter scripts through the plugin's batched meter mode (WFHipMeterGroup: levels read one frame late), against the
    plugin's own CPU class; judged like the batch fuzz (within tolerance of the reference, or no farther from the exactly summed
    level than the reference itself is)"""
    import os
    from pathlib import Path
    import test_golden as tg
    from oracle import wfref
    from helpers import assert_levels_close
    os.environ["WF_HIP_LIBRARY"] = str(Path(__file__).resolve().parent.parent / "waveform_amd" / "libwaveform_hip.so")
    os.environ["WF_HIP_BATCHED"] = "1"
    cfg_dict, steps = draw_meter(seed)
    cfg = scenarios.make_config(cfg_dict)
    sc = dict(cfg=cfg_dict, steps=steps, record="all")
    before = wfref.hip_fallback_ticks()
    late = tg._OneFrameLate(scenarios.RefBackend(cfg, isa="hip"))
    assert late.be.src.using_hip
    scenarios.play(late, sc)
    got = late.finish()
    assert late.be.src.using_hip and wfref.hip_fallback_ticks() == before, "fell back to the CPU class"
    want = scenarios.play(scenarios.RefBackend(cfg, isa="generic"), sc)
    truth = scenarios.play(scenarios.OracleBackend(cfg, exact=True), sc)
    assert len(got) == len(want) == len(truth)
    for t, (g, w, x) in enumerate(zip(got, want, truth)):
        what = f"batched drop-in meter case {seed} tick {t} ({cfg_dict})"
        assert g["silent"] == w["silent"] or g["silent"] == x["silent"], what + f": m_last_silent {g['silent']} != {w['silent']}"
        assert_levels_close(g["db"], w["db"], x["db"], what + " levels")


def run_dropin_batched_wave_case(seed):
    """waveform scripts through the plugin's batched waveform mode (rows read one frame late), against the plugin's own CPU
    class.  Volume normalisation is taken out of the drawn configuration: such sources stay synchronous by design."""
    import os
    from pathlib import Path
    import test_golden as tg
    from oracle import wfref
    from helpers import assert_db_close
    os.environ["WF_HIP_LIBRARY"] = str(Path(__file__).resolve().parent.parent / "waveform_amd" / "libwaveform_hip.so")
    os.environ["WF_HIP_BATCHED"] = "1"
    cfg_dict, steps, sync_ms = draw_wave(seed)
    cfg_dict = {k: v for k, v in cfg_dict.items() if k not in ("normalize_volume", "volume_target", "max_gain")}
    cfg = scenarios.make_config(cfg_dict)
    sc = dict(cfg=cfg_dict, steps=steps, record="all", sync_ms=sync_ms)
    before = wfref.hip_fallback_ticks()
    late = tg._OneFrameLate(scenarios.RefBackend(cfg, isa="hip"))
    assert late.be.src.using_hip
    scenarios.play(late, sc)
    got = late.finish()
    assert late.be.src.using_hip and wfref.hip_fallback_ticks() == before, "fell back to the CPU class"
    want = scenarios.play(scenarios.RefBackend(cfg, isa="generic"), sc)
    assert len(got) == len(want), (len(got), len(want))
    for t, (g, w) in enumerate(zip(got, want)):
        what = f"batched drop-in wave case {seed} tick {t} ({cfg_dict}, sync {sync_ms} ms)"
        assert g["silent"] == w["silent"], what + f": m_last_silent {g['silent']} != {w['silent']}"
        assert_db_close(g["db"], w["db"], what + " rows", lin_eps=None)


DROPIN_SEEDS = {"pow2": range(0, 80), "any": range(0, 40), "meter": range(0, 60), "wave": range(0, 60)}


@pytest.mark.gpu
@pytest.mark.parametrize("family,seed", [(k, s) for k, r in DROPIN_SEEDS.items() for s in r])
def test_reference_plugin_with_hip_tick_on_random_case(family, seed):
    """consecutive seeds of every family through oracle/_ref's WAVSource with WAVSourceHIP as the tick implementation
    (synchronous mode), against the plugin's own CPU class in the same process"""
    from oracle import wfref
    if not wfref.available():
        pytest.skip("oracle/_ref/libwfref.so not built")
    run_dropin_case(seed, family)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(0, 60))
def test_reference_plugin_with_batched_hip_meter_on_random_case(seed):
    """the batched meter mode of the plugin on consecutive seeds of the meter family"""
    from oracle import wfref
    if not wfref.available():
        pytest.skip("oracle/_ref/libwfref.so not built")
    run_dropin_batched_meter_case(seed)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(0, 60))
def test_reference_plugin_with_batched_hip_waveform_on_random_case(seed):
    """the batched waveform mode of the plugin on consecutive seeds of the wave family"""
    from oracle import wfref
    if not wfref.available():
        pytest.skip("oracle/_ref/libwfref.so not built")
    run_dropin_batched_wave_case(seed)


@pytest.mark.gpu
@pytest.mark.parametrize("family,seed", [(k, s) for k, r in (("pow2", range(0, 60)), ("any", range(0, 30))) for s in r])
def test_reference_plugin_with_batched_hip_tick_on_random_case(family, seed):
    """the batched plugin mode (sources share a handle, rows read one frame late) on consecutive seeds"""
    from oracle import wfref
    if not wfref.available():
        pytest.skip("oracle/_ref/libwfref.so not built")
    run_dropin_batched_case(seed, family)


@pytest.mark.gpu
def test_zz_display_arm_stays_rare():
    """runs last in this module: the second arm of the display check (bars / curve held against the render of the device's
    own rows) may be taken by at most 5 of 1000 display checks, and every configuration the draws produce must have been
    accepted -- WF_HIP_ERR_UNSUPPORTED is legal only for the corners include/wf_hip.h documents, none of which a draw reaches"""
    import helpers
    st = helpers.ARM_STATS
    print(f"rows: {st['values']} dB values in {st['calls']} comparisons; decided by the linear arm {st['linear_arm']} "
          f"({st['linear_arm'] / max(st['values'], 1):.2e}), of them above {helpers.VISIBLE_DB} dB {st['linear_arm_visible']} "
          f"({st['linear_arm_visible'] / max(st['values'], 1):.2e}); below {helpers.DEEP_DB} dB and lower than the reference: {st['deep']}")
    print(f"display checks {ARM['display_checks']}, second arm taken {ARM['display_arm']}: {ARM['arm_cases'][:10]}")
    print(f"unsupported configurations: {len(ARM['unsupported'])}: {ARM['unsupported'][:10]}")
    if st["values"] >= 10_000_000:  # the whole module ran in this process
        assert st["linear_arm"] <= 1e-5 * st["values"] and st["linear_arm_visible"] <= 1e-6 * st["values"], st
    if ARM["display_checks"] >= 200:
        assert ARM["display_arm"] <= max(1, 5 * ARM["display_checks"] // 1000), ARM["arm_cases"][:20]
    assert not ARM["unsupported"], ARM["unsupported"][:20]
