"""CPU-side unit tests (no GPU): synthetic-audio hash, host tables of the product vs the oracle,
the kernel's phase functions run in the wavefront emulator vs the oracle, C-ABI export list."""
import sys
import ctypes as C
import re
import subprocess
from pathlib import Path

import numpy as np
import pytest

import scenarios
import emu_binding as emu
from helpers import assert_db_close
from oracle import restate
from tools import synth

ROOT = Path(__file__).resolve().parent.parent


# ---- wf_synth.h -----------------------------------------------------------------------------------
def test_synth_numpy_matches_c():
    from oracle import wfref
    if not wfref.available():
        pytest.skip("libwfref.so not built")
    L = wfref.lib()
    for stream, ch, i0 in ((0, 0, 0), (7, 1, 12345), (65535, 1, 2**33 + 5)):
        a = synth.noise(synth.DEFAULT_SEED, stream, ch, i0, 257)
        b = np.array([L.wfref_noise(synth.DEFAULT_SEED, stream, ch, i0 + i) for i in range(257)], np.float32)
        assert np.array_equal(a, b)
    x = synth.noise(1, 2, 0, 0, 1 << 16)
    assert x.min() >= -1.0 and x.max() < 1.0 and abs(float(x.mean())) < 0.01
    assert abs(float(x.var()) - 1 / 3) < 0.01


# ---- host tables: product code (wf_host_tables.cpp) must be bit-identical to the oracle --------------
TABLE_CFGS = [
    dict(fft_size=4096, stereo=1, slope=1.0),
    dict(fft_size=1024, window=2, slope=0.37, rolloff_q=2.0, rolloff_rate=6.5, cutoff_low=120, cutoff_high=9000),
    dict(fft_size=2048, window=3, bars=1, interp_mode=1),
    dict(fft_size=8192, window=4, bars=1, interp_mode=2, log_scale=0, bar_width=3, bar_gap=1, width=1280),
    dict(fft_size=16384, window=5, sine_exponent=4, tsmoothing=2, gravity=0.9, bars=1, interp_mode=0, mirror_freq_axis=1),
    dict(fft_size=2048, window=0, tsmoothing=0, bars=1, interp_mode=1, stereo=1, channel_spacing=8, min_bar_height=4, rounded_caps=1),
]


@pytest.mark.parametrize("ov", TABLE_CFGS)
def test_host_tables_bit_identical_to_oracle(ov):
    cfg = scenarios.make_config(ov)
    o = restate.OracleSource(cfg)
    w, wsum = o.window()
    scal = emu.host_table(cfg, 6, 1 / 75)
    assert scal[0] == np.float32(wsum)
    assert scal[1] == np.float32(o.gravity(1 / 75))
    assert scal[2] == np.float32(restate.db_min())
    assert int(scal[3]) == o.num_bars
    for which, want in ((0, w), (1, o.slope()), (2, o.rolloff())):
        got = emu.host_table(cfg, which)
        if want is None:
            assert got.size == 0
        else:
            assert np.array_equal(got, want), f"table {which}"
    if cfg.bars:
        assert np.array_equal(emu.host_table(cfg, 3), o.interp_indices())
        assert np.array_equal(emu.host_table(cfg, 5).astype(np.int32), o.band_widths())
        kw, radius, taps = o.interp_weights()
        got = emu.host_table(cfg, 4)
        if kw is None:
            assert got.size == 0
        else:
            assert np.array_equal(got, kw) and int(scal[4]) == radius and int(scal[5]) == taps


def test_unsupported_fft_sizes_are_rejected():
    for n in (64, 65552, 131072, 3000):  # below the reference's minimum, above its maximum (65536), not a multiple of 16
        cfg = scenarios.make_config(dict(fft_size=n))
        with pytest.raises(ValueError):
            emu.host_table(cfg, 0)
    for n in (128, 800, 10912, 10928, 12000, 16400, 32768, 48000, 65520, 65536):  # every multiple of 16 in [128, 65536] is taken
        emu.host_table(scenarios.make_config(dict(fft_size=n)), 0)


def test_degenerate_display_configurations_are_rejected():
    """configurations whose setup would divide by zero (the reference would produce NaNs) come back as errors"""
    for bad in (dict(bars=1, bar_width=0, bar_gap=0), dict(bars=1, width=0), dict(curve=1, width=1),
                dict(bars=1, floor_db=0, ceiling_db=0), dict(curve=1, filter_mode=1, filter_radius=float("nan")),
                dict(bars=1, filter_mode=7)):
        cfg = scenarios.make_config(dict(fft_size=1024, **bad))
        with pytest.raises(ValueError):
            emu.host_table(cfg, 0)


# ---- kernel phases in the wavefront emulator vs the oracle --------------------------------------------
@pytest.mark.parametrize("n", [512, 1024, 2048, 4096, 8192, 16384, 32768])
@pytest.mark.parametrize("hop", [800, 441])
def test_emulated_kernel_matches_oracle(n, hop):
    cfg = scenarios.make_config(dict(fft_size=n, stereo=1, slope=1.0, fast_peaks=1))
    o = restate.OracleSource(cfg)
    ticks = 4
    cap = 1
    while cap < n + ticks * hop:
        cap *= 2
    ring = np.zeros((2, cap), np.float32)
    ts = np.zeros((2, n // 2), np.float32)
    w = n
    for t in range(ticks):
        a = synth.block(99, 0, 1, 2, t * hop, hop)[0]
        ring[:, w:w + hop] = a
        w += hop
        o.feed_and_tick(a)
        db, stats = emu.tick(cfg, ring, w, ts)
        assert_db_close(db[0], o.decibels(), f"N={n} hop={hop} tick {t}")
        for c in range(2):
            want = o.tsmooth(c)
            assert np.allclose(ts[c], want, rtol=2e-5, atol=1e-12)


def test_emulated_kernel_window_delay():
    """the A/V-sync delay: the window ends `delay` frames before the newest sample"""
    n, hop = 2048, 800
    cfg = scenarios.make_config(dict(fft_size=n, stereo=1, tsmoothing=0))
    audio = synth.block(5, 0, 1, 2, 0, 4 * hop)[0]
    ring = np.zeros((2, 8192), np.float32)
    ring[:, n:n + 4 * hop] = audio
    ts = np.zeros((2, n // 2), np.float32)
    o = restate.OracleSource(cfg)
    for t in range(4):
        o.feed_and_tick(audio[:, t * hop:(t + 1) * hop])
        db, _ = emu.tick(cfg, ring, n + 4 * hop, ts, delay=(3 - t) * hop)
        assert_db_close(db[0], o.decibels(), f"delay tick {t}")


def test_lds_budget_and_conflicts():
    """LDS per spectrum stays within the occupancy plan and every exchange access is bank-conflict free under the
    lane-group model of MI355X_MICROARCH.md (reads and writes, all geometries)"""
    budget = {512: 5 * 1024, 1024: 5 * 1024, 2048: 9 * 1024, 4096: 17408, 8192: 34 * 1024, 16384: 68 * 1024, 32768: 133 * 1024}
    for n, b in budget.items():
        assert 0 < emu.lib().wfemu_lds_bytes(n) <= b
        cfg = scenarios.make_config(dict(fft_size=n, stereo=1))
        ring = np.ascontiguousarray(synth.block(1, 0, 1, 2, 0, 2 * n)[0], np.float32)
        ts = np.zeros((2, n // 2), np.float32)
        _, st = emu.tick(cfg, ring, 2 * n, ts)
        rd_instr, rd_ideal, rd_actual, wr_instr, wr_ideal, wr_actual = st
        assert rd_instr > 0 and wr_instr > 0
        assert wr_actual == wr_ideal, (n, "LDS writes must be bank-conflict free", wr_ideal, wr_actual)
        assert rd_actual == rd_ideal, (n, "LDS reads must be bank-conflict free", rd_ideal, rd_actual)


def test_bar_segment_tables_are_a_permutation_of_the_flat_tables():
    """wf::bar_segments (what the kernel's bars phase reads: per thread 4 * blocks consecutive bins from a base that is a
    multiple of 4, so that the row is read as 16-byte LDS words) must weigh the bins exactly as the flat table does, bar by
    bar: evaluate both forms on a random dB row in float64"""
    rng = np.random.default_rng(7)
    threads = {1024: 64, 2048: 64, 4096: 128, 8192: 256, 16384: 512}
    points = {1024: 8, 2048: 16, 4096: 16, 8192: 16, 16384: 16}
    seen = 0
    for n in (1024, 2048, 4096, 8192, 16384):
        for mode, extra in ((1, {}), (2, dict(log_scale=0)), (0, dict(mirror_freq_axis=1, bar_width=10, bar_gap=2))):
            cfg = scenarios.make_config(dict(fft_size=n, stereo=1, bars=1, interp_mode=mode, **extra))
            coef = emu.host_table(cfg, 7).astype(np.float64)
            bins = emu.host_table(cfg, 8).astype(np.int64)
            off = emu.host_table(cfg, 9).astype(np.int64)
            T, mb = threads[n], points[n] // 4 + 1
            lc = emu.bar_lanes(cfg, T, mb, 0)
            if lc is None:  # too many bars / segments too long for the registers: the kernel takes its chunked path
                continue
            seen += 1
            base = emu.bar_lanes(cfg, T, mb, 1).astype(np.int64)
            seg = emu.bar_lanes(cfg, T, mb, 2).astype(np.int64)
            num_segs, blocks = (int(v) for v in emu.bar_lanes(cfg, T, mb, 3))
            assert num_segs <= T and 1 <= blocks <= mb and len(seg) == len(off) and len(base) == T
            lc = lc.astype(np.float64).reshape(blocks, T, 4)
            assert np.all(lc[:, num_segs:, :] == 0)
            assert np.all(base % 4 == 0) and np.all(base >= 0) and np.all(base + 4 * blocks <= n // 2)
            db = rng.uniform(-120.0, 0.0, n // 2)
            lb = base[None, :, None] + 4 * np.arange(blocks)[:, None, None] + np.arange(4)[None, None, :]
            per_thread = (lc * db[lb]).sum(axis=(0, 2))          # one partial per thread / segment
            for b in range(len(off) - 1):
                flat = float((coef[off[b]:off[b + 1]] * db[bins[off[b]:off[b + 1]]]).sum())
                lanes = float(per_thread[seg[b]:seg[b + 1]].sum())
                assert abs(flat - lanes) <= 1e-9 * max(1.0, abs(flat)), (n, mode, b, flat, lanes)
    assert seen >= 10
    # more bars than threads per spectrum: no segment form, the kernel takes its chunked path
    many = scenarios.make_config(dict(fft_size=1024, stereo=1, bars=1, interp_mode=1, width=1920, bar_width=1, bar_gap=0, log_scale=0))
    assert emu.bar_lanes(many, 64, 3, 0) is None


def test_slope_factors_formed_on_the_device_stay_next_to_the_reference_table():
    """Policy<G>::SLOPE_LINEAR (wf_tick_phases.hpp): the geometries from 2048 samples form m_slope_modifiers[k] as
    fma(k, 3 * slope / (M - 1), 1) instead of loading the reference's table log10f(10 * powf(1000, k * slope / (M - 1)))
    (src/source.cpp:1283-1290; bit-identical in wf_host_tables.cpp).  The two differ by the table's own powf / log10f roundings:
    pinned here at 5e-7 relative over the reference's slider range, 1/20 of the parity tolerance."""
    worst = 0.0
    for n in (2048, 4096, 8192, 16384, 32768):
        for slope in (0.05, 0.3, 1.0, 2.25, 5.0):
            cfg = scenarios.make_config(dict(fft_size=n, stereo=1, slope=slope))
            tab = emu.host_table(cfg, 1).astype(np.float64)
            M = n // 2
            assert len(tab) == M
            step = np.float32(3.0 * float(np.float32(slope)) / (M - 1))
            k = np.arange(M, dtype=np.float32)
            dev = (k.astype(np.float64) * np.float64(step) + 1.0).astype(np.float32).astype(np.float64)   # one fma: a single rounding
            worst = max(worst, float(np.max(np.abs(dev - tab) / tab)))
    assert worst <= 5e-7, worst


def test_bar_piece_tables_replayed_lane_by_lane():
    """wf::bar_pieces (the wave-private form of the bars tail: no barrier between parking the row and reading it, partial sums
    added by a six-step DPP scan, the last wavefront to arrive adds the pieces).  Replays what the kernel does with the tables on
    a random dB row -- per-thread dot products over 16-byte words, the segmented prefix scan step by step exactly as the DPP
    controls move lanes (row_shr:1/2/4/8 inside rows of 16, row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3), the
    slots, the sum per bar -- against the flat table; and checks the property the missing barrier rests on: a thread reads
    only bins its own wavefront has written."""
    rng = np.random.default_rng(11)
    geoms = {512: (64, 4), 1024: (64, 8), 2048: (64, 16), 4096: (128, 16), 8192: (256, 16), 16384: (512, 16), 32768: (512, 32)}
    seen = 0
    for n, (T, P) in geoms.items():
        for mode, extra in ((1, {}), (2, dict(log_scale=0)), (0, dict(mirror_freq_axis=1, bar_width=10, bar_gap=2)), (1, dict(width=1200)), (0, dict(width=1000, bar_width=20, bar_gap=5, log_scale=0))):
            cfg = scenarios.make_config(dict(fft_size=n, stereo=1, bars=1, interp_mode=mode, **extra))
            coef = emu.host_table(cfg, 7).astype(np.float64)
            bins = emu.host_table(cfg, 8).astype(np.int64)
            off = emu.host_table(cfg, 9).astype(np.int64)
            mb = P // 4 + 2
            lc = emu.bar_pieces(cfg, T, P, mb, 0)
            if lc is None:
                # (at most 64 bars -- lane b of the last wavefront finishes bar b -- and at most 64 segments per wavefront: on a
                # log axis most bars lie in the first 256 bins, i.e. in wavefront 0; bar_segments' layouts take the rest)
                assert len(off) - 1 > 30, (n, mode, extra, "the form should exist")
                continue
            seen += 1
            base = emu.bar_pieces(cfg, T, P, mb, 1).astype(np.int64)
            info = emu.bar_pieces(cfg, T, P, mb, 2).astype(np.int64)
            piece = emu.bar_pieces(cfg, T, P, mb, 3).astype(np.int64)
            num_segs, blocks, num_slots = (int(v) for v in emu.bar_pieces(cfg, T, P, mb, 4))
            M, wps = n // 2, T // 64
            assert 1 <= blocks <= mb and len(base) == T and len(info) == T and len(piece) == len(off) and num_segs <= T
            lc = lc.astype(np.float64).reshape(blocks, T, 4)
            assert np.all(base % 4 == 0) and np.all(base >= 0) and np.all(base + 4 * blocks <= M)
            # wave-private reads: bin k is parked by the wavefront that owns chunk k // 256, i.e. wave (k // 256) % wps
            lb = base[None, :, None] + 4 * np.arange(blocks)[:, None, None] + np.arange(4)[None, None, :]   # [blocks][T][4]
            owner = (lb // 256) % wps
            wave = (np.arange(T) // 64)[None, :, None]
            used = lc != 0
            assert np.all(owner[used] == wave.repeat(blocks, 0).repeat(4, 2)[used]), (n, mode, "a thread weighs a bin another wavefront parks")
            if wps > 1:  # stricter: the 16-byte words it READS lie in its own wavefront's chunks too (garbage times 0 would be NaN)
                segs = (info != 0) | (np.abs(lc).sum(axis=(0, 2)) > 0)
                assert np.all((owner == wave)[:, segs, :]), (n, mode, "a segment reads bins of another wavefront")
            db = rng.uniform(-120.0, 0.0, M)
            part = (lc * db[lb]).sum(axis=(0, 2))
            # the scan, wavefront by wavefront
            tot = part.copy()
            for w in range(wps):
                v = tot[64 * w:64 * w + 64].copy()
                f = info[64 * w:64 * w + 64]
                lane = np.arange(64)
                for bit, d in enumerate((1, 2, 4, 8)):
                    src = np.where((lane & 15) >= d, np.roll(v, d), 0.0)          # row_shr:d, bound_ctrl: 0 from outside the row
                    v = v + np.where((f >> bit) & 1, src, 0.0)
                src = np.where(((lane >> 4) & 1) == 1, v[np.maximum((lane & ~15) - 1, 0)], 0.0)   # row_bcast:15, rows 1 and 3
                v = v + np.where((f >> 4) & 1, src, 0.0)
                src = np.where(lane >= 32, v[31], 0.0)                                           # row_bcast:31, rows 2 and 3
                v = v + np.where((f >> 5) & 1, src, 0.0)
                tot[64 * w:64 * w + 64] = v
            slots = np.full(max(num_slots, len(off)), np.nan)
            ends = np.nonzero(info >> 8)[0]
            slot_of = (info[ends] >> 8) - 1
            assert len(set(slot_of.tolist())) == len(slot_of), "two lanes write one slot"
            slots[slot_of] = tot[ends]
            for b in range(len(off) - 1):
                flat = float((coef[off[b]:off[b + 1]] * db[bins[off[b]:off[b + 1]]]).sum())
                got = float(slots[b]) if wps == 1 else float(slots[piece[b]:piece[b + 1]].sum())
                if off[b + 1] == off[b]:
                    got = 0.0 if wps == 1 and np.isnan(got) else got
                assert abs(flat - got) <= 1e-9 * max(1.0, abs(flat)), (n, mode, extra, b, flat, got)
    assert seen >= 20


def _replay_bar_ps(tab, M, T, P, db):
    """what spectrum_tick_kernel does with wf::bar_ps' lane table (ps_park / ps_finish, wf_tick_phases.hpp), lane by lane: the
    parked row with its guard zeros, the thread-major group sums, the float64 prefix over 16-bin quads, one look-up and one 7-tap
    window per lane (two lanes per sub-band), the swap between the lanes of a pair, the segmented scan over a bar's lanes.
    Returns {bar: sum / count} in float64 arithmetic on the float32 table values (the device's float32 roundings are the GPU
    suite's business)."""
    tab = tab.reshape(3, 64, 4)
    NG = P // 4
    row = np.zeros(M + 8)
    row[4:4 + M] = db
    grp = db.reshape(M // 4, 4).astype(np.float32)
    own = ((grp[:, 0] + grp[:, 1]) + (grp[:, 2] + grp[:, 3]))   # float32 four-bin sums, as the kernel forms them
    gs = np.zeros(T * NG + 4 * NG, np.float32)
    for g in range(M // 4):
        gs[(g % T) * NG + g // T] = own[g]

    def quad_groups(Q):
        g0 = 4 * Q
        return (g0 & (T - 1)) * NG + g0 // T
    QUADS = M // 16
    NQ = (QUADS + 63) // 64
    qp = np.zeros(QUADS + 1)
    run_tot = 0.0
    for ll in range(64):                      # serial per lane + the scan over the lanes = a running total
        for k in range(NQ):
            Q = ll * NQ + k
            if Q < QUADS:
                p = quad_groups(Q)
                qt = np.float32(np.float32(gs[p] + gs[p + NG]) + np.float32(gs[p + 2 * NG] + gs[p + 3 * NG]))
                qp[Q] = run_tot
                run_tot += float(qt)
    qp[QUADS] = run_tot
    f, e, sw = np.zeros(64), np.zeros(64), np.zeros(64)
    info = np.zeros(64, np.int64)
    for lane in range(64):
        w = tab[:, lane, :].astype(np.float32)
        c = np.concatenate([w[0], w[1][:3]]).astype(np.float64)
        sw[lane] = float(w[1][3])
        q, inf = (int(x) for x in w[2][:2].view(np.uint32))
        info[lane] = inf
        assert 0 <= q <= M
        win = row[q + 1:q + 8]
        x = min(q + 4, M)
        ng, nb = (x >> 2) & 3, x & 3
        p = quad_groups(x >> 4)
        f[lane] = qp[x >> 4] + sum(float(gs[p + i * NG]) for i in range(ng)) + sum(win[6 - j] for j in range(nb))
        e[lane] = float(c @ win)
    lane = np.arange(64)
    other = lane ^ 1
    v = np.where(lane & 1, 0.0, sw * (f[other] - f) + e + e[other])
    for bit, d in enumerate((1, 2, 4, 8)):
        src = np.where((lane & 15) >= d, np.roll(v, d), 0.0)
        v = v + np.where((info >> bit) & 1, src, 0.0)
    src = np.where(((lane >> 4) & 1) == 1, v[np.maximum((lane & ~15) - 1, 0)], 0.0)
    v = v + np.where((info >> 4) & 1, src, 0.0)
    src = np.where(lane >= 32, v[31], 0.0)
    v = v + np.where((info >> 5) & 1, src, 0.0)
    out = {}
    for l in range(64):
        bar = ((int(info[l]) >> 8) & 0xff) - 1
        if bar >= 0:
            assert bar not in out and l % 2 == 0, "two lanes finish one bar / a high-edge lane finishes one"
            out[bar] = v[l] / (int(info[l]) >> 16)
    return out


@pytest.mark.parametrize("n,c", [(48016, 8), (17488, 8), (65424, 8), (33824, 8), (48064, 32), (48064, 16), (17728, 32), (65344, 32), (33472, 16)])
def test_bluestein_rows_tables_replayed(n, c):
    """The sizes above 16384 whose n/2 has a large prime factor: n/2 = C R, the column step and C rows of R points by chirp-z over
    L >= 2 R - 1 points (big_br_columns_kernel / big_br_rows_kernel, wf_big.hpp).  The host tables (wf::build_bluestein_rows)
    replayed with numpy's FFT as the container transform must deliver the n/2-point DFT -- the packed real transform the
    reference's fftwf_plan_dft_r2c_1d computes (src/source.cpp:423-431) -- row k1 holding Z[k1 + C k2]."""
    from tests import emu_binding as emu
    points = n // 2
    R = points // c
    L, rowtw, bhat, q = emu.bluestein_rows(points, c)
    assert L >= 2 * R - 1 and L & (L - 1) == 0 and (L < 4 * R or L == 1024) and L in (1024, 2048, 4096, 8192)
    rng = np.random.default_rng(n)
    z = (rng.standard_normal(points) + 1j * rng.standard_normal(points)).astype(np.complex64)
    want = np.fft.fft(z.astype(np.complex128))
    cols = z.reshape(c, R)                                   # cols[cc, n2] = z[n2 + R cc]
    wc = np.exp(-2j * np.pi * np.outer(np.arange(c), np.arange(c)) / c)
    a = (wc @ cols.astype(np.complex128)) * rowtw            # a[k1, n2], chirp and column twiddle folded in
    y = np.zeros((c, L), np.complex128)
    y[:, :R] = a
    # the kernel's form: R = FFT(conj(FFT(y) bhat)) (= L conj(y (*) chirp)), Z_row = q conj(R)
    r2 = np.fft.fft(np.conj(np.fft.fft(y, axis=1) * bhat), axis=1)[:, :R]
    rows = q * np.conj(r2)
    got = np.empty(points, np.complex128)
    for k1 in range(c):
        got[k1::c] = rows[k1]
    err = np.abs(got - want).max() / np.abs(want).max()
    assert err < 2e-6, err


def test_bar_prefix_sum_tables_replayed_lane_by_lane():
    """wf::bar_ps (round 5: the bar reduction as two look-ups into a float64 prefix sum of the row + two 7-tap edge windows per
    sub-band of identical weight rows, reference src/filter.hpp:194-211 / src/source.cpp:876-884) against the flat per-bin
    coefficient table on random rows -- white-noise-like and with 150 dB of dynamic range --, every geometry that runs it,
    Lanczos / Catmull-Rom / point, log and linear axes, mirrored axis."""
    rng = np.random.default_rng(23)
    geoms = {512: (64, 4), 1024: (64, 8), 2048: (64, 16), 4096: (128, 16), 8192: (256, 16), 16384: (512, 16), 32768: (512, 32)}
    seen = 0
    for n, (T, P) in geoms.items():
        for mode, extra in ((1, {}), (2, {}), (2, dict(log_scale=0)), (1, dict(log_scale=0)), (0, {}), (0, dict(mirror_freq_axis=1, bar_width=10, bar_gap=2)),
                            (1, dict(width=1200)), (1, dict(mirror_freq_axis=1)), (1, dict(width=1920, bar_width=5, bar_gap=1)),
                            (1, dict(cutoff_low=20, cutoff_high=24000)), (2, dict(cutoff_low=0, cutoff_high=300)), (1, dict(width=500, bar_width=40, bar_gap=10))):
            cfg = scenarios.make_config(dict(fft_size=n, stereo=1, bars=1, interp_mode=mode, **extra))
            coef = emu.host_table(cfg, 7).astype(np.float64)
            bins = emu.host_table(cfg, 8).astype(np.int64)
            off = emu.host_table(cfg, 9).astype(np.int64)
            widths = emu.host_table(cfg, 5).astype(np.int64)
            tab = emu.bar_ps(cfg, T, 0)
            if tab is None:
                assert len(off) - 1 > 24, (n, mode, extra, "the form should exist")   # (at most 32 sub-bands: one wavefront, two lanes each)
                continue
            num_lanes, num_subs = (int(v) for v in emu.bar_ps(cfg, T, 1))
            assert num_lanes == 64 and len(off) - 1 <= num_subs <= 32 and tab.size == 64 * 12
            seen += 1
            M = n // 2
            for kind in range(2):
                db = (rng.uniform(-90.0, -20.0, M) if kind == 0 else np.where(rng.uniform(size=M) < 0.02, -3.0, -150.0)).astype(np.float32).astype(np.float64)
                got = _replay_bar_ps(tab, M, T, P, db)
                assert sorted(got) == list(range(len(off) - 1)), (n, mode, extra)
                for b in range(len(off) - 1):
                    flat = float((coef[off[b]:off[b + 1]] * db[bins[off[b]:off[b + 1]]]).sum()) / widths[b]
                    assert abs(flat - got[b]) <= 3e-7 * max(1.0, abs(flat)), (n, mode, extra, kind, b, flat, got[b])
    assert seen >= 35, seen


def test_power_of_two_kernels_do_not_spill():
    """the fused kernels of the power-of-two sizes (the measured ones) must fit their register budget: a few bytes of scratch
    per lane cost N = 1024 ten percent in round 2 before anyone looked.  Compiles each geometry on its own (hipcc
    cross-compiles without a GPU) and reads the compiler's resource remarks."""
    import concurrent.futures as cf
    src = ROOT / "waveform_amd" / "csrc"

    def usage(n):
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-fno-slp-vectorize", f"-I{ROOT / 'include'}",
               f"-I{src}", f"-DWF_TU_GEOM={n}", "-Rpass-analysis=kernel-resource-usage", "-c", str(src / "wf_tick_geom.hip"), "-o", "/dev/null"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        out, name, vgprs = [], None, 0
        for line in r.stderr.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                name = m.group(1)
            m = re.search(r" VGPRs: (\d+)", line)
            if m:
                vgprs = int(m.group(1))
            m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
            if m and name and "spectrum_tick_kernel" in name:
                out.append((name, int(m.group(1)), vgprs))
        return out

    with cf.ThreadPoolExecutor(6) as ex:
        results = list(ex.map(usage, (512, 1024, 2048, 4096, 8192, 16384, 32768)))
    seen = seen_mixed = seen_small = 0
    for res in results:
        for name, scratch, vgprs in res:
            # wf::Variant{spw, aligned, split, dec, tlds, blu, both, mr, mrs, mir, disp, plan} as the mangled name spells it: the
            # members in order, trailing zeros left out
            m = re.search(r"XtlNS_7VariantE((?:L[ib]\d+E)*)E+vNS_8TickArgsE$", name)
            assert m, name
            vals = [int(x) for x in re.findall(r"L[ib](\d+)E", m.group(1))]
            vals += [0] * (12 - len(vals))
            blu, mixed, small = vals[5] == 1, vals[7] == 1, vals[8] == 1
            if blu and not mixed:
                continue  # Bluestein: the compatibility path, a few spills tolerated (bounded by the next test)
            if small:     # the one-wavefront container's small-radix instantiation: five waves per SIMD, 96 registers, three words parked
                seen_small += 1
                assert scratch <= 16 and vgprs <= 96, f"{name}: {scratch} B of scratch per lane, {vgprs} VGPRs"
                continue
            if mixed:     # the mixed-radix transform inside the same instantiation (wf_mixed.hpp): no scratch either
                seen_mixed += 1
                assert scratch == 0 and vgprs <= 128, f"{name}: {scratch} B of scratch per lane, {vgprs} VGPRs"
                continue
            seen += 1
            assert scratch == 0, f"{name} uses {scratch} bytes of scratch per lane"
            # four waves per SIMD: one VGPR over 128 cost N = 2048 a quarter of its occupancy (and 5-15 %) in round 2
            # (N = 32768: 512 threads of 32 points, one workgroup per CU by its LDS = two waves per SIMD with 256 registers each)
            limit = 256 if "GeomILi32768ELi512E" in name else 128
            assert vgprs <= limit, f"{name} needs {vgprs} VGPRs (limit {limit}): a wave per SIMD fewer"
    assert seen >= 12 and seen_mixed >= 6 and seen_small >= 1


def test_compatibility_path_kernels_stay_near_their_register_budget(tmp_path):
    """The Bluestein instantiations (two transforms and a chirped fetch of six registers per point in one kernel) do not all fit
    128 registers: up to 28 B per lane in scratch (none on the 16384- and 32768-sample geometries since the display's per-thread
    words are fetched where they are used).  Tolerated on that path -- but bounded
    here, so that a change that pushes a kernel into a kilobyte of scratch (the fused 65536 kernel that was abandoned had 1-2 KB)
    is seen.  The 65536-sample kernel (both rows and the end of the tick in one workgroup of 512 threads at 256 registers) sits at
    its register limit: 20 B per lane today, stored once at the start and read back once at the end; its predecessor spilled
    196 B per lane inside the fetch until the sums were parked in the exchange buffer -- with one workgroup per CU that was 420 MB
    of device-memory traffic per launch -- and two attempts at requesting the epilogue's operands earlier tipped this one into
    800 B (EXPERIMENTS.md, round 4)."""
    src = ROOT / "waveform_amd" / "csrc"
    tu = tmp_path / "compat.hip"
    tu.write_text('''#include <hip/hip_runtime.h>
#include "wf_hip.h"
#include "wf_host_tables.hpp"
#include "wf_kernels.hpp"
#include "wf_big.hpp"
template __global__ void wf::spectrum_tick_kernel<wf::G2048, wf::Variant{.spw = 2, .blu = true}>(wf::TickArgs);
template __global__ void wf::spectrum_tick_kernel<wf::G4096, wf::Variant{.spw = 2, .blu = true}>(wf::TickArgs);
template __global__ void wf::spectrum_tick_kernel<wf::G16384, wf::Variant{.spw = 1, .split = true, .blu = true}>(wf::TickArgs);
template __global__ void wf::big_whole_kernel<true>(wf::TickArgs);
template __global__ void wf::big_whole_kernel<false>(wf::TickArgs);
template __global__ void wf::big_epilogue_kernel<1>(wf::TickArgs);
''')
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-fno-slp-vectorize", f"-I{ROOT / 'include'}", f"-I{src}",
           "-Rpass-analysis=kernel-resource-usage", "-c", str(tu), "-o", "/dev/null"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    name, seen = None, {}
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name and ("spectrum_tick_kernel" in name or "big_whole_kernel" in name or "big_epilogue_kernelILi1" in name):
            seen[name] = int(m.group(1))
    assert len(seen) == 6, seen
    for name, scratch in seen.items():
        limit = 0 if ("big_epilogue" in name or "GeomILi16384ELi512E" in name) else 32
        assert scratch <= limit, f"{name}: {scratch} B of scratch per lane (limit {limit})"


def test_mixed_radix_plans_and_transforms(tmp_path):
    """wf_mixed.hpp (fft sizes with small prime factors and at most one prime factor of 17 .. 127, computed directly instead of by Bluestein) on the host: tests/emu/mr_check.cpp runs
    every in-register DFT against the definition, checks the plan of EVERY multiple of 16 up to 16384 (exists exactly for the
    smooth sizes, multiplies to n / 2, respects the kernel's constraints), and replays the passes lane by lane against a double
    DFT for a spread of sizes"""
    exe = tmp_path / "mr_check"
    src = ROOT / "waveform_amd" / "csrc"
    cmd = ["g++", "-std=c++20", "-O1", f"-I{ROOT / 'include'}", f"-I{src}", str(ROOT / "tests" / "emu" / "mr_check.cpp"), str(src / "wf_host_tables.cpp"), "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and run.stdout.strip().endswith("ok") and "planned 624 sizes" in run.stdout, run.stdout[-3000:]


# ---- C ABI -----------------------------------------------------------------------------------------------
def _declared_functions(header: Path):
    text = re.sub(r"/\*.*?\*/", "", header.read_text(), flags=re.S)
    text = re.sub(r"//.*", "", text)
    text = re.sub(r"#ifdef WF_DEV_BUILD.*?#endif", "", text, flags=re.S)  # the test aids: development builds only (checked below)
    names = set()
    for m in re.finditer(r"\b(wf_[a-z0-9_]+)\s*\(", text):
        names.add(m.group(1))
    return names


def test_library_exports_every_declared_symbol():
    so = ROOT / "waveform_amd" / "libwaveform_hip.so"
    assert so.exists(), "build the library first: make -C waveform_amd/csrc"
    declared = _declared_functions(ROOT / "include" / "wf_hip.h") | _declared_functions(ROOT / "include" / "wf_config.h")
    declared -= {"wf_synth_mix64", "wf_synth_key", "wf_synth_sample", "wf_synth_noise"}
    nm = subprocess.run(["nm", "-D", "--defined-only", str(so)], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in nm.splitlines() if " T " in line}
    missing = sorted(declared - exported)
    assert not missing, f"declared in include/*.h but not exported: {missing}"
    L = C.CDLL(str(so))  # loads without a GPU
    for name in declared:
        assert hasattr(L, name)
    L.wf_hip_abi_version.restype = C.c_int
    assert L.wf_hip_abi_version() == 13


def test_release_library_carries_no_laboratory():
    """The release library exports no test aid and at most 75 entry points (the boundary is a thin C ABI: the reference's operator
    interface is four virtuals, src/source.hpp:273-277); the development build (same kernel objects, -DWF_DEV_BUILD on the two
    host-side translation units) has the two hooks the tests need.  Every measurement switch in the kernels (WF_EXP_*: kernels that
    end early or skip a phase, with WRONG results) defaults to 0 and is refused by wf_dev_guard.hpp outside development builds."""
    def exported(path):
        nm = subprocess.run(["nm", "-D", "--defined-only", str(path)], capture_output=True, text=True, check=True).stdout
        return {line.split()[-1] for line in nm.splitlines() if " T " in line and line.split()[-1].startswith("wf_hip_")}
    rel = exported(ROOT / "waveform_amd" / "libwaveform_hip.so")
    dev = exported(ROOT / "waveform_amd" / "libwaveform_hip_dev.so")
    assert not [n for n in rel if "debug" in n], [n for n in rel if "debug" in n]
    assert len(rel) <= 75, (len(rel), sorted(rel))
    assert dev - rel == {"wf_hip_debug_age", "wf_hip_multi_debug_fail_next_gather"}, sorted(dev - rel)
    csrc = ROOT / "waveform_amd" / "csrc"
    guard = (csrc / "wf_dev_guard.hpp").read_text()
    seen = set()
    for p in list(csrc.glob("*.hpp")) + list(csrc.glob("*.hip")) + list(csrc.glob("*.cpp")):
        if p.name == "wf_dev_guard.hpp":
            continue
        txt = p.read_text()
        seen |= set(re.findall(r"\bWF_EXP_[A-Z0-9_]+\b", txt))
        for name, val in re.findall(r"#define\s+(WF_EXP_[A-Z0-9_]+)\s+(\S+)", txt):
            assert val == "0", f"{p.name}: {name} defaults to {val}"
        if "WF_PHASE_TIMING" in txt or "WF_EXP_" in txt:
            assert "wf_dev_guard.hpp" in txt or p.name in ("wf_tick_phases.hpp", "wf_big.hpp", "wf_hip.hip", "wf_hip_plan.hip"), p.name  # (those include it through wf_hip_internal.hpp / wf_kernels.hpp)
    assert seen <= {"WF_EXP_NO_TAIL", "WF_EXP_CUT_AT"}, f"measurement switches in the product kernels: {sorted(seen)}"
    for name in seen:
        assert name in guard, f"{name} is not refused by wf_dev_guard.hpp outside development builds"
    # and the guard works: a release compile with a stray -D stops
    src = "#define WF_EXP_NO_TAIL 1\n#include \"wf_dev_guard.hpp\"\nint x;\n"
    r = subprocess.run(["g++", "-fsyntax-only", "-x", "c++", "-I", str(csrc), "-"], input=src, capture_output=True, text=True)
    assert r.returncode != 0 and "WF_DEV_BUILD" in r.stderr
    r = subprocess.run(["g++", "-fsyntax-only", "-x", "c++", "-DWF_DEV_BUILD", "-I", str(csrc), "-"], input=src, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_no_device_fails_loudly():
    import waveform_amd as wf
    if wf.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(wf.WfHipError) as e:
        wf.SpectrumBatch(wf.Config.defaults(), 4)
    assert e.value.code == -3  # WF_HIP_ERR_NO_DEVICE: no CPU fallback exists


def test_product_does_not_reference_the_oracle():
    """the product tree must not import, include or link anything under oracle/"""
    bad = []
    for p in (ROOT / "waveform_amd").rglob("*"):
        if p.suffix in (".py", ".hpp", ".cpp", ".hip", ".h") or p.name == "Makefile":
            txt = p.read_text(errors="ignore")
            if re.search(r"oracle[/.]|wforacle|wfref|wfemu", txt) and p.name not in ("wf_fft_core.hpp", "wf_tick_phases.hpp"):
                bad.append(str(p))
    assert not bad, bad
    ldd = subprocess.run(["ldd", str(ROOT / "waveform_amd" / "libwaveform_hip.so")], capture_output=True, text=True).stdout
    assert "wforacle" not in ldd and "wfref" not in ldd and "fftw" not in ldd
    assert "rccl" not in ldd, "librccl.so is dlopen()ed by the multi-device group, never linked"


def test_multi_device_group_fails_loudly_without_a_device():
    import waveform_amd as wf
    if wf.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(wf.WfHipError) as e:
        wf.MultiBatch(wf.Config.defaults(bars=1), 8, [0, 1])
    assert e.value.code == -3  # WF_HIP_ERR_NO_DEVICE
    with pytest.raises(wf.WfHipError) as e:
        wf.MultiBatch(wf.Config.defaults(bars=1), 8, [])
    assert e.value.code == -1  # WF_HIP_ERR_INVALID before any device is looked at


# ---- configuration checks that run before any device is touched ---------------------------------------------------------
def _create_code(**overrides):
    """status of wf_hip_create for defaults + overrides on a machine without a GPU: -3 (NO_DEVICE) means the configuration
    itself was accepted, -1 / -2 that it was rejected as invalid / unsupported"""
    import waveform_amd as wf
    if wf.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(wf.WfHipError) as e:
        wf.SpectrumBatch(wf.Config.defaults(**overrides), 2)
    return e.value.code


def test_fft_sizes_accepted_and_rejected():
    # every multiple of 16 in [128, 65536] -- the reference's own range (src/source.cpp:349, :359-363, :562-565) -- is taken
    for n in list(range(128, 65537, 16 * 97)) + [256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 800, 10912, 10928, 16400, 48000, 65520]:
        assert _create_code(fft_size=n) == -3, n
    for n in (64, 808, 65552, 131072):  # below the minimum, not a multiple of 16, above the maximum
        assert _create_code(fft_size=n) == -2, n


def test_meter_and_waveform_configurations():
    assert _create_code(meter=1) == -3
    assert _create_code(meter=1, meter_ms=0) == -1
    assert _create_code(meter=1, ceiling_db=-70) == -3       # ceiling <= floor is repaired as get_settings does (src/source.cpp:572-576): 0 / -120
    assert _create_code(meter=1, fft_size=12345) == -3       # fft_size is ignored in meter mode (it becomes the buffer length)
    assert _create_code(waveform=1) == -3
    assert _create_code(waveform=1, width=0) == -1
    assert _create_code(waveform=1, meter_ms=0) == -1
    assert _create_code(waveform=1, width=9000) == -2        # more points per row than the kernel stages in LDS


def test_headers_are_plain_c_and_the_example_links(tmp_path):
    """include/*.h compile as C99 with no warnings, and the C example links against the library and fails loudly
    (WF_HIP_ERR_NO_DEVICE) where there is no GPU"""
    import os
    import waveform_amd as wf
    for name in ("batch_spectrum", "multi_gpu_bars"):
        exe = tmp_path / name
        cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", f"-I{ROOT / 'include'}", str(ROOT / "examples" / f"{name}.c"),
               f"-L{ROOT / 'waveform_amd'}", "-lwaveform_hip", "-lm", "-o", str(exe)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        if wf.device_count() > 0:
            continue  # a GPU is present: the example would run (tests/test_gpu_multi.py runs multi_gpu_bars there)
        env = dict(os.environ, LD_LIBRARY_PATH=f"{ROOT / 'waveform_amd'}:/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
        run = subprocess.run([str(exe)], capture_output=True, text=True, env=env)
        assert run.returncode == 1 and "-3" in run.stderr, (name, run.returncode, run.stderr)


def test_no_fusable_rounding_intrinsics_in_device_code():
    """HIP compiles device code with -ffp-contract=fast and treats __fmul_rn / __fadd_rn as plain operators: a product fed
    into a sum is one fma, not two roundings (the level meter's smoothing was 7e-4 dB off on a tick where the sum cancels).
    Arithmetic that must round like the reference's scalar code goes through mul_unfused / add_unfused / meter_ema
    (#pragma clang fp contract(off)); the intrinsics may only appear where nothing can fuse with them."""
    import re
    root = Path(__file__).resolve().parent.parent / "waveform_amd" / "csrc"
    bad = []
    for f in sorted(root.glob("*.h*")):
        for n, line in enumerate(f.read_text().splitlines(), 1):
            code = line.split("//")[0]
            if re.search(r"__f(add|sub)_rn\s*\([^;]*__fmul_rn|__fmul_rn\s*\([^;]*__f(add|sub)_rn", code):
                bad.append(f"{f.name}:{n}: {line.strip()}")
    assert not bad, "products feeding sums through rounding intrinsics (they fuse):\n" + "\n".join(bad)


# ---- evidence guards and the node check's verdicts ---------------------------------------------------
BENCH_SHAPE_KEYS = None


def _bench_shape_keys():
    """the profile keys bench.py's line looks up: the headline's and shape_list()'s last column"""
    global BENCH_SHAPE_KEYS
    if BENCH_SHAPE_KEYS is None:
        sys.path.insert(0, str(ROOT))
        import bench
        import waveform_amd as wf
        BENCH_SHAPE_KEYS = ["cfg3_n4096"] + [s[5] for s in bench.shape_list(wf)]
    return BENCH_SHAPE_KEYS


def test_every_bench_shape_has_a_committed_profile_of_the_newest_round():
    """bench.py replays roofline.traffic from profiles/rNN*_<shape>_pmc.json (PMC counters need rocprofv3 passes of their own).  A
    shape without a summary -- or with one from an older evidence set than the others -- silently reports traffic null in the
    driver's line: every shape bench.py names must have a summary, all of the newest set, each with a kernel name, HBM bytes per
    launch and a tick span.  (That the `kernel` strings are the kernels that RUN is checked where there is a device:
    tests/test_gpu_fullsize.py::test_committed_profiles_are_of_the_kernels_that_run.)"""
    import json
    keys = _bench_shape_keys()
    tags = sorted({p.name.split("_")[0] for k in keys for p in (ROOT / "profiles").glob(f"r*_{k}_pmc.json")})
    assert tags, "no committed rocprofv3 summaries"
    newest = tags[-1]
    for k in keys:
        p = ROOT / "profiles" / f"{newest}_{k}_pmc.json"
        assert p.exists(), f"{k}: no summary in the newest evidence set ({newest}); bench.py would replay an older one or report traffic null"
        d = json.loads(p.read_text())
        assert d.get("kernel") and d.get("hbm_bytes_per_launch", 0) > 0 and (d.get("trace") or {}).get("tick_span_ns", 0) > 0, (k, {x: d.get(x) for x in ("kernel", "hbm_bytes_per_launch")})


def test_bench_line_puts_the_baseline_shapes_first():
    """the driver keeps the head of the standard keys and the last 8 KB of stdout: BASELINE shapes first, the reference-range extras last"""
    keys = _bench_shape_keys()
    assert keys[:4] == ["cfg3_n4096", "cfg4_n16384_bars", "cfg5shape_8192streams_barsonly", "cfg2_batch"] and keys[-2:] == ["n65536", "n800_mixed_radix"], keys


def test_node_check_names_every_reason():
    """tools/node_check.py exits non-zero with ONE LINE PER REASON: RCCL refuses the device list, peer access denied on a pair,
    a device more than 10 % slower than the best, a wrong gathered copy -- and none on a clean node"""
    sys.path.insert(0, str(ROOT))
    from tools import node_check
    ok_run = {"leg": "default", "transport_asked": "default", "transport": "rccl", "transport_note": "", "devices": [0, 1, 2, 3], "verified": True,
              "devices_with_a_wrong_copy": [], "ms_per_tick_without_gather": {"max": 0.111, "per_device": [0.110, 0.111, 0.109, 0.110]}}
    peers = [{"from": a, "to": b, "can_access": True, "enable_rc": 0, "ok": True} for a in range(4) for b in range(4) if a != b]
    assert node_check.reasons({"peer_access": peers, "runs": [ok_run]}, 4) == []
    # RCCL refused: the default leg fell back to peer copies, the library's text says why
    r = dict(ok_run, transport="peer", transport_note="ncclCommInitAll failed: unhandled system error")
    why = node_check.reasons({"peer_access": peers, "runs": [r]}, 4)
    assert len(why) == 1 and "RCCL refuses the device list [0, 1, 2, 3]" in why[0] and "ncclCommInitAll failed" in why[0]
    # ... or the leg that asked for it by name failed outright
    r = {"leg": "rccl, one channel", "transport_asked": "rccl", "devices": [0, 1, 2, 3], "error": "WF_HIP_MULTI_TRANSPORT=rccl: librccl.so not loadable: x", "verified": False}
    why = node_check.reasons({"peer_access": peers, "runs": [r]}, 4)
    assert len(why) == 1 and "RCCL refuses the device list" in why[0] and "librccl.so not loadable" in why[0]
    # peer access denied on one ordered pair
    bad = [dict(p, can_access=False, enable_rc=-1, ok=False) if (p["from"], p["to"]) == (2, 3) else p for p in peers]
    why = node_check.reasons({"peer_access": bad, "runs": [ok_run]}, 4)
    assert len(why) == 1 and "device 2 cannot address device 3" in why[0]
    # one device 15 % slower than the best
    r = dict(ok_run, ms_per_tick_without_gather={"max": 0.1265, "per_device": [0.110, 0.1265, 0.109, 0.110]})
    why = node_check.reasons({"peer_access": peers, "runs": [r]}, 4)
    assert len(why) == 1 and "device 1 takes 126.5 us per tick, 16 % more than the best device (109.0 us)" in why[0], why
    # two shards on ONE device (the 1-GPU rehearsal of the peer legs) share it: no spread verdict there
    r = dict(ok_run, devices=[0, 0], ms_per_tick_without_gather={"max": 0.2, "per_device": [0.1, 0.2]})
    assert node_check.reasons({"peer_access": [], "runs": [r]}, 1) == []
    # a wrong copy; a host-fed device that failed
    r = dict(ok_run, devices_with_a_wrong_copy=[3], verified=False)
    why = node_check.reasons({"peer_access": peers, "runs": [r], "host_fed": {"verified": False, "per_device": [{"device": 1, "error": "hipHostMalloc failed"}]}}, 4)
    assert len(why) == 2 and "device indices [3]" in why[0] and "host-fed leg: device 1: hipHostMalloc failed" in why[1]
