"""Golden-vector parity.

tests/golden/*.npz hold what the reference itself (oracle/_ref: phandasm/waveform's own TUs +
vendored FFTW, generic path) produced for every scenario of tests/scenarios.py.
  * CPU  (-m "not gpu"): the oracle restatement must reproduce them  -> pins the oracle
  * GPU  (-m gpu)      : libwaveform_hip.so, through its C ABI, must reproduce them

Tolerances: dB values 1e-5 relative (BASELINE.json north_star) + 1e-4 dB absolute floor; bar tops
(pixels) 1e-5 relative + 2e-3 px (the reference's own FMA3 and scalar bar paths differ by 1e-4 dB);
m_last_silent must match exactly.
"""
import json
from pathlib import Path

import numpy as np
import pytest

import scenarios
from helpers import assert_db_close

GOLDEN = Path(__file__).resolve().parent / "golden"
NAMES = sorted(scenarios.SCENARIOS)


def _load(name):
    p = GOLDEN / f"{name}.npz"
    assert p.exists(), f"{p} missing: run tools/make_golden.py where /root/reference exists"
    z = np.load(p)
    meta = json.loads(bytes(z["meta"]).decode())
    assert meta["cfg"] == scenarios.SCENARIOS[name]["cfg"], "fixture is stale: regenerate with tools/make_golden.py"
    return z, meta


def _check(name, backend):
    sc = scenarios.SCENARIOS[name]
    z, meta = _load(name)
    recs = scenarios.play(backend, sc)
    assert len(recs) == meta["n_ticks"]
    silent = np.array([r["silent"] for r in recs], np.uint8)
    assert np.array_equal(silent, z["silent"]), f"{name}: m_last_silent sequence {silent} != reference {z['silent']}"
    if "rms" in z.files:
        # m_input_rms: the reference adds its 48000 squares one by one in float; 1e-5 relative is the north star's tolerance
        got = np.array([r["rms"] for r in recs], np.float64)
        assert np.all(np.abs(got - z["rms"]) <= 1e-5 * np.abs(z["rms"]) + 1e-9), f"{name}: m_input_rms {got} vs reference {z['rms']}"
    for t, r in scenarios.recorded(recs, sc["record"]):
        assert_db_close(r["db"], z[f"db_{t}"], f"{name} tick {t} decibels", deep=True)
        if f"bars_{t}" in z.files:
            got, want = r["bars"], z[f"bars_{t}"]
            assert got is not None
            err = np.abs(got.astype(np.float64) - want)
            assert np.all(err <= 1e-5 * np.abs(want) + 2e-3), f"{name} tick {t} bars: max err {err.max():.3e} px"
        c = 0
        while f"verts_{t}_c{c}" in z.files:
            # the vertex buffer at that channel's gs_draw: as many vertices, x bit for bit (integer products and the same float
            # additions), y as close as the bars are, z = w = 0
            got, want = r["verts"][c], z[f"verts_{t}_c{c}"]
            assert got.shape == want.shape, f"{name} tick {t} channel {c}: {got.shape} vertices vs the reference's {want.shape}"
            assert np.array_equal(got[..., 0], want[..., 0]), f"{name} tick {t} channel {c}: vertex x coordinates differ"
            err = np.abs(got[..., 1].astype(np.float64) - want[..., 1])
            assert np.all(err <= 1e-5 * np.abs(want[..., 1]) + 2e-3), f"{name} tick {t} channel {c} vertex y: max err {err.max():.3e} px"
            assert not got[..., 2:].any() and not want[..., 2:].any()
            c += 1


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_reference(name):
    cfg = scenarios.make_config(scenarios.SCENARIOS[name]["cfg"])
    _check(name, scenarios.OracleBackend(cfg))


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_reproduces_reference(name):
    cfg = scenarios.make_config(scenarios.SCENARIOS[name]["cfg"])
    be = scenarios.HipBackend(cfg)
    try:
        _check(name, be)
    finally:
        be.close()


# ---- the drop-in: the reference plugin itself, with only the per-tick DSP virtuals replaced by the HIP binding -------------
SPECTRUM_DROPIN = ["underflow_normalize_sync_2048", "cfg1_mono_1024", "cfg2_stereo_2048_nosmooth", "cfg3_stereo_4096_ema_slope", "cfg4_16384_tv_lanczos_bars",
                   "mono_mix_4096_tv_fastpeaks", "rolloff_catrom_linear", "ragged_hops", "silence_cycle", "half_silent_stereo",
                   "hide_show", "muted_packets", "timeout_spectrum", "split_8192_half_silent", "small_512_stereo_bars",
                   "small_128_single_dup_curve", "large_32768_single_tv", "any_800_mono_mix_bars", "any_4160_stereo_silence",
                   "sync_spectrum_2048", "sync_spectrum_4096_normalize_mono", "normalize_4096_stereo", "normalize_mono_muted_ragged",
                   "normalize_long_1024", "curve_4096_lanczos_gauss", "plugin_defaults_4096", "curve_4096_catrom_wide_gauss",
                   "huge_65536_stereo_bars", "huge_65536_mono_mix_tv_curve", "any_48000_single_dup"]
DROPIN = SPECTRUM_DROPIN + [
    # WAVSourceHIP::tick_meter
    "meter_rms_stereo", "meter_peak_mono_tv_fastpeaks", "meter_nosmooth_ragged", "meter_silence_cycle", "meter_half_silent",
    "meter_hide_show_timeout",
    # WAVSourceHIP::tick_waveform
    "wave_stereo_800", "wave_mono_mix_ragged", "wave_single_dup_stall", "wave_hide_timeout_sync", "wave_sync_burst", "wave_normalize",
    "wave_tick_before_audio_sync"]


def _hip_env(batched):
    import os
    from oracle import wfref
    if not wfref.available():
        pytest.skip("oracle/_ref/libwfref.so not built")
    os.environ["WF_HIP_LIBRARY"] = str(Path(__file__).resolve().parent.parent / "waveform_amd" / "libwaveform_hip.so")
    os.environ["WF_HIP_BATCHED"] = "1" if batched else "0"
    return wfref


@pytest.mark.gpu
@pytest.mark.parametrize("name", DROPIN)
def test_reference_plugin_with_hip_tick(name):
    """oracle/_ref's WAVSource (update / capture_audio / tick / render_bars run verbatim) with WAVSourceHIP
    (host/wav_source_hip.cpp) as the tick_spectrum / tick_meter / tick_waveform implementation, against the same golden
    vectors; synchronous mode (WF_HIP_BATCHED=0: one handle per source, results inside the call).  The device path must
    still be the one in use when the scenario ends, and no tick may have been served by the CPU class."""
    wfref = _hip_env(batched=False)
    cfg = scenarios.make_config(scenarios.SCENARIOS[name]["cfg"])
    before = wfref.hip_fallback_ticks()
    drawn, on_host = wfref.hip_device_renders(), wfref.hip_host_renders()
    be = scenarios.RefBackend(cfg, isa="hip")
    assert be.src.using_hip, "WAVSourceHIP fell back to the CPU path: the HIP library did not load or no gfx950 device"
    _check(name, be)
    assert be.src.using_hip, "WAVSourceHIP released the device path during the scenario"
    assert wfref.hip_fallback_ticks() == before, "ticks were served by the reference's CPU class"
    _check_renders(wfref, cfg, drawn, on_host)


def _check_renders(wfref, cfg, drawn, on_host):
    """WAVSourceHIP::render (host/wav_source_hip.cpp): every render() of a spectrum display must have been drawn from the device's
    bar tops / curve points and vertices -- none handed to the reference's own render_bars / render_curve, i.e. no
    apply_interp_filter*, apply_filter* or vertex loop on the host (the bars and vertex buffers the scenario compares are then
    the device's)"""
    if cfg.meter or cfg.waveform or not (cfg.bars or cfg.curve) or scenarios.no_vertex_buffer(cfg):
        return
    assert wfref.hip_host_renders() == on_host, "render() ran the reference's interpolation and vertex loops on the host"
    assert wfref.hip_device_renders() > drawn, "no render() was served from the device's display"


@pytest.mark.gpu
@pytest.mark.parametrize("batched", [False, True])
@pytest.mark.parametrize("shader", ["gradient", "pulse:peak_magnitude", "pulse:peak_frequency"])
def test_mirrored_axis_shader_constants_come_from_the_device(shader, batched):
    """render modes whose shader constants follow the row's smallest y -- gradient (grad_height) and pulse (color_base, by
    magnitude or by position) -- on a MIRRORED frequency axis: the reference takes miny / minpos before the mirror image
    replaces the upper half of the row (src/source.cpp:1548-1567, :1411-1424).  WAVSourceHIP::render finds them from the
    device's mirrored row plus the one value the outputs above the middle had before (WF_HIP_OUT_PREMIRROR): every render must
    be served from the device (host_renders unchanged) and hand set_shader_vars what the plugin's own CPU class hands it --
    bars with an odd and an even count, stereo and mono, and a curve."""
    wfref = _hip_env(batched=batched)
    mode, _, pulse = shader.partition(":")
    extra = dict(render_mode=mode, grad_ratio=repr(0.75), color_base=0xFF102030, color_crest=0xFFE0D0C0)
    if pulse:
        extra["pulse_mode"] = pulse
    names = {"gradient": ["grad_height", "grad_center", "grad_offset"], "pulse": ["color_base"]}[mode]
    layouts = [dict(fft_size=2048, stereo=1, bars=1, interp_mode=1, mirror_freq_axis=1, vertices=1),                        # 26 bars
               dict(fft_size=4096, stereo=0, bars=1, interp_mode=2, mirror_freq_axis=1, vertices=1, width=810, slope=1.0),  # 27 bars
               dict(fft_size=1024, stereo=1, bars=1, interp_mode=0, mirror_freq_axis=1, vertices=1, width=640, bar_width=9, bar_gap=2, log_scale=0),
               dict(fft_size=2048, stereo=1, curve=1, interp_mode=2, mirror_freq_axis=1, vertices=1, width=801)]
    steps = [("noise", 800), ("tick",)] * 3 + [("noise_amp", 800, 0.02), ("tick",)] * 3 + [("noise_ch0_only", 800), ("tick",)] * 2
    for cfg_dict in layouts:
        cfg = scenarios.make_config(cfg_dict)
        sc = dict(cfg=cfg_dict, steps=steps, record="all")

        class Shaded:
            """a RefBackend whose observe() also records the shader constants of that frame's render"""
            def __init__(self, isa):
                self.be = scenarios.RefBackend(cfg, isa=isa, extra_settings=extra)
                self.capture_channels = self.be.capture_channels
            def __getattr__(self, k):
                return getattr(self.be, k)
            def observe(self):
                rec = self.be.observe()
                rec["shader"] = {n: self.be.src.shader_value(n) for n in names}
                return rec
        before, drawn, on_host = wfref.hip_fallback_ticks(), wfref.hip_device_renders(), wfref.hip_host_renders()
        hip = Shaded("hip")
        assert hip.be.src.using_hip
        if batched:
            late = _OneFrameLate(hip)
            scenarios.play(late, sc)
            got = late.finish()
        else:
            got = scenarios.play(hip, sc)
        assert hip.be.src.using_hip and wfref.hip_fallback_ticks() == before
        assert wfref.hip_host_renders() == on_host, f"{cfg_dict}: render() went back to the host loops for a mirrored axis with {shader}"
        assert wfref.hip_device_renders() > drawn
        want = scenarios.play(Shaded("generic"), sc)
        assert len(got) == len(want)
        for t, (g, w) in enumerate(zip(got, want)):
            for n in names:
                assert w["shader"][n] is not None and g["shader"][n] is not None, (cfg_dict, shader, n)
                d = np.abs(g["shader"][n].astype(np.float64) - w["shader"][n])
                assert np.all(d <= 1e-5 * np.abs(w["shader"][n]) + 2e-3), f"{cfg_dict} {shader} tick {t}: shader constant {n} {g['shader'][n]} != {w['shader'][n]}"


@pytest.mark.gpu
@pytest.mark.parametrize("batched", [False, True])
@pytest.mark.parametrize("shader", ["gradient", "pulse:peak_frequency"])
def test_mirrored_filtered_axis_keeps_the_reference_loops_for_miny(shader, batched):
    """The same render modes with the Gaussian filter ON (apply_filter runs before the miny / minpos loop,
    src/source.cpp:1541-1547): the pre-mirror outputs next to the middle are blends of the clamped top value with lower bars, the
    minimum may sit at any of them, and one kept value cannot stand for them -- WAVSourceHIP::render hands those frames to the
    reference's own loops (host renders counted) and the shader constants equal the plugin's CPU class.  Without gradient / pulse
    the same filtered, mirrored display is drawn from the device."""
    wfref = _hip_env(batched=batched)
    mode, _, pulse = shader.partition(":")
    extra = dict(render_mode=mode, grad_ratio=repr(0.75), color_base=0xFF102030, color_crest=0xFFE0D0C0)
    if pulse:
        extra["pulse_mode"] = pulse
    names = {"gradient": ["grad_height", "grad_center", "grad_offset"], "pulse": ["color_base"]}[mode]
    layouts = [dict(fft_size=2048, stereo=1, bars=1, interp_mode=1, mirror_freq_axis=1, vertices=1, filter_mode=1, filter_radius=2.5),
               dict(fft_size=2048, stereo=0, curve=1, interp_mode=2, mirror_freq_axis=1, vertices=1, width=401, filter_mode=1, filter_radius=4.0)]
    # loud low end, quiet top: the smallest y of the filtered row sits right above the middle, among the blended outputs
    steps = [("noise", 800), ("tick",)] * 3 + [("noise_amp", 800, 0.02), ("tick",)] * 3
    for cfg_dict in layouts:
        cfg = scenarios.make_config(cfg_dict)
        sc = dict(cfg=cfg_dict, steps=steps, record="all")

        class Shaded:
            def __init__(self, isa, extra_settings):
                self.be = scenarios.RefBackend(cfg, isa=isa, extra_settings=extra_settings)
                self.capture_channels = self.be.capture_channels
            def __getattr__(self, k):
                return getattr(self.be, k)
            def observe(self):
                rec = self.be.observe()
                rec["shader"] = {n: self.be.src.shader_value(n) for n in names}
                return rec
        on_host = wfref.hip_host_renders()
        hip = Shaded("hip", extra)
        assert hip.be.src.using_hip
        if batched:
            late = _OneFrameLate(hip)
            scenarios.play(late, sc)
            got = late.finish()
        else:
            got = scenarios.play(hip, sc)
        assert hip.be.src.using_hip
        assert wfref.hip_host_renders() > on_host, f"{cfg_dict} {shader}: a filtered, mirrored row's miny was taken from the device's single pre-mirror value"
        want = scenarios.play(Shaded("generic", extra), sc)
        assert len(got) == len(want)
        for t, (g, w) in enumerate(zip(got, want)):
            for n in names:
                d = np.abs(g["shader"][n].astype(np.float64) - w["shader"][n])
                assert np.all(d <= 1e-5 * np.abs(w["shader"][n]) + 2e-3), f"{cfg_dict} {shader} tick {t}: shader constant {n} {g['shader'][n]} != {w['shader'][n]}"


class _OneFrameLate:
    """plays a scenario on a backend whose outputs lag one video frame (the batched plugin mode): every tick's record is
    taken at the following tick; one extra tick at the end collects the last frame"""

    def __init__(self, backend):
        self.be = backend
        self.capture_channels = backend.capture_channels
        self.pending = False
        self.records = []

    def __getattr__(self, name):
        return getattr(self.be, name)

    def tick(self, seconds):
        self.be.tick(seconds)
        if self.pending:
            self.records.append(self.be.observe())
        self.pending = True

    def observe(self):
        return None  # records are taken one tick later, see tick()

    def finish(self):
        self.be.tick(1.0 / 60.0)
        self.records.append(self.be.observe())
        return self.records


@pytest.mark.gpu
@pytest.mark.parametrize("name", SPECTRUM_DROPIN)
def test_reference_plugin_with_batched_hip_tick(name):
    """The plugin mode that can win (SURVEY.md section 7 "Drop-in latency"): sources of one configuration share a batch,
    one tick per video frame for all of them, every source reads its row ONE FRAME LATER.  The reference plugin with
    WAVSourceHIP in batched mode reproduces every spectrum golden scenario shifted by exactly one tick."""
    wfref = _hip_env(batched=True)
    sc = scenarios.SCENARIOS[name]
    cfg = scenarios.make_config(sc["cfg"])
    z, meta = _load(name)
    before, rms_before = wfref.hip_fallback_ticks(), wfref.hip_host_rms_updates()
    drawn, on_host = wfref.hip_device_renders(), wfref.hip_host_renders()
    late = _OneFrameLate(scenarios.RefBackend(cfg, isa="hip"))
    assert late.be.src.using_hip
    scenarios.play(late, sc)
    recs = late.finish()
    assert late.be.src.using_hip and wfref.hip_fallback_ticks() == before
    assert wfref.hip_host_rms_updates() == rms_before, "update_input_rms ran on the host: the device RMS producer was not in use"
    _check_renders(wfref, cfg, drawn, on_host)
    assert len(recs) == meta["n_ticks"]
    silent = np.array([r["silent"] for r in recs], np.uint8)
    assert np.array_equal(silent, z["silent"]), f"{name}: m_last_silent sequence {silent} != reference {z['silent']} (one frame late)"
    if "rms" in z.files:
        # m_input_rms as the device's update_input_rms left it at that batch's tick (WAVSourceHIP::update_input_rms feeds the
        # squared peaks; the sum is a tree, the reference adds its 48000 squares one by one in float)
        got = np.array([r["rms"] for r in recs], np.float64)
        assert np.all(np.abs(got - z["rms"]) <= 1e-5 * np.abs(z["rms"]) + 1e-9), f"{name}: m_input_rms {got} vs reference {z['rms']} (one frame late)"
    for t, r in scenarios.recorded(recs, sc["record"]):
        assert_db_close(r["db"], z[f"db_{t}"], f"{name} tick {t} decibels, read one frame later", deep=True)
        if f"bars_{t}" in z.files:
            err = np.abs(r["bars"].astype(np.float64) - z[f"bars_{t}"])
            assert np.all(err <= 1e-5 * np.abs(z[f"bars_{t}"]) + 2e-3), f"{name} tick {t} bars: max err {err.max():.3e} px"
        c = 0
        while f"verts_{t}_c{c}" in z.files:  # the vertex buffer at that channel's gs_draw, filled from the device's vertices one frame later
            got, want = r["verts"][c], z[f"verts_{t}_c{c}"]
            assert got.shape == want.shape and np.array_equal(got[..., 0], want[..., 0]), f"{name} tick {t} channel {c}: vertex count / x"
            err = np.abs(got[..., 1].astype(np.float64) - want[..., 1])
            assert np.all(err <= 1e-5 * np.abs(want[..., 1]) + 2e-3), f"{name} tick {t} channel {c} vertex y: max err {err.max():.3e} px"
            c += 1


METER_DROPIN = ["meter_rms_stereo", "meter_peak_mono_tv_fastpeaks", "meter_nosmooth_ragged", "meter_silence_cycle", "meter_half_silent",
                "meter_hide_show_timeout"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", METER_DROPIN)
def test_reference_plugin_with_batched_hip_meter(name):
    """The level meter in the plugin's batched mode (WFHipMeterGroup: sources of one meter configuration share a handle, one
    meter_tick_kernel per video frame, levels read one frame later): every meter golden scenario, shifted by exactly one tick."""
    wfref = _hip_env(batched=True)
    sc = scenarios.SCENARIOS[name]
    cfg = scenarios.make_config(sc["cfg"])
    z, meta = _load(name)
    before = wfref.hip_fallback_ticks()
    late = _OneFrameLate(scenarios.RefBackend(cfg, isa="hip"))
    assert late.be.src.using_hip
    scenarios.play(late, sc)
    recs = late.finish()
    assert late.be.src.using_hip and wfref.hip_fallback_ticks() == before
    assert len(recs) == meta["n_ticks"]
    silent = np.array([r["silent"] for r in recs], np.uint8)
    assert np.array_equal(silent, z["silent"]), f"{name}: m_last_silent sequence {silent} != reference {z['silent']} (one frame late)"
    for t, r in scenarios.recorded(recs, sc["record"]):
        assert_db_close(r["db"], z[f"db_{t}"], f"{name} tick {t} levels, read one frame later", lin_eps=None)
        if f"bars_{t}" in z.files:
            err = np.abs(r["bars"].astype(np.float64) - z[f"bars_{t}"])
            assert np.all(err <= 1e-5 * np.abs(z[f"bars_{t}"]) + 2e-3), f"{name} tick {t} bars: max err {err.max():.3e} px"


WAVE_BATCHED = ["wave_stereo_800", "wave_mono_mix_ragged", "wave_single_dup_stall", "wave_hide_timeout_sync", "wave_sync_burst",
                "wave_tick_before_audio_sync"]  # (wave_normalize: sources with volume normalisation stay synchronous)


@pytest.mark.gpu
@pytest.mark.parametrize("name", WAVE_BATCHED)
def test_reference_plugin_with_batched_hip_waveform(name):
    """The waveform display in the plugin's batched mode (WFHipMeterGroup with cfg.waveform: sources of one configuration share
    a handle; every frame one ragged ingest of what each tick_waveform consumed, every member's A/V-sync reserve and audio
    timestamp, one waveform_tick_kernel, rows read one frame later): every waveform golden scenario, shifted by exactly one tick."""
    wfref = _hip_env(batched=True)
    sc = scenarios.SCENARIOS[name]
    cfg = scenarios.make_config(sc["cfg"])
    z, meta = _load(name)
    before = wfref.hip_fallback_ticks()
    late = _OneFrameLate(scenarios.RefBackend(cfg, isa="hip"))
    assert late.be.src.using_hip
    scenarios.play(late, sc)
    recs = late.finish()
    assert late.be.src.using_hip and wfref.hip_fallback_ticks() == before
    assert len(recs) == meta["n_ticks"]
    silent = np.array([r["silent"] for r in recs], np.uint8)
    assert np.array_equal(silent, z["silent"]), f"{name}: m_last_silent sequence {silent} != reference {z['silent']} (one frame late)"
    for t, r in scenarios.recorded(recs, sc["record"]):
        assert_db_close(r["db"], z[f"db_{t}"], f"{name} tick {t} rows, read one frame later", lin_eps=None)


@pytest.mark.gpu
def test_waveform_sources_share_one_batch():
    """24 waveform sources of one configuration: one handle, one ragged ingest + one waveform_tick_kernel + one readback per
    video frame.  Every source has its own audio, amplitude and packet sizes, some hide, stall or lose their capture along the way;
    each one's rows at frame t+1 are what the reference's own CPU class has at frame t.  Then the cost per source and frame
    against the synchronous device path and the reference's AVX class in the same harness."""
    import os
    from tools import synth
    wfref = _hip_env(batched=True)
    os.environ["WF_HIP_BATCH_CAPACITY"] = "64"
    cfg_dict = dict(waveform=1, stereo=1, width=800, meter_ms=150)
    cfg = scenarios.make_config(cfg_dict)
    n_src, frames = 24, 40
    before = wfref.hip_fallback_ticks()
    srcs = [scenarios.RefBackend(cfg, isa="hip") for _ in range(n_src)]
    refs = [scenarios.RefBackend(cfg, isa="generic") for _ in range(n_src)]
    assert all(s.src.using_hip for s in srcs)
    want_prev = [None] * n_src
    pos = [0] * n_src
    for f in range(frames):
        for i, (s, o) in enumerate(zip(srcs, refs)):
            stalled = (i % 7 == 3) and f in (25, 26)          # no packet and no tick in these frames: the stream is paused
            if f == 20 and i % 5 == 1:
                s.set_hidden(True), o.set_hidden(True)
            if f == 24 and i % 5 == 1:
                s.set_hidden(False), o.set_hidden(False)
            if stalled:
                continue
            if i % 11 == 4 and f == 30:
                s.timeout(), o.timeout()
            else:
                hop = (800, 441, 1024, 960)[i % 4] if f % 3 != 2 or i % 2 else 0   # some frames bring no packet at all
                if hop:
                    a = synth.block(scenarios.SEED, 300 + i, 1, 2, pos[i], hop)[0] * np.float32(1.0 if i % 3 else 0.05)
                    pos[i] += hop
                    for b in (s, o):
                        b.push(a, muted=False)
            for b in (s, o):
                b.tick(1.0 / 60.0)
            got = s.observe()
            if want_prev[i] is not None:
                w = want_prev[i]
                assert got["silent"] == w["silent"], f"source {i} frame {f}: m_last_silent"
                assert_db_close(got["db"], w["db"], f"source {i} frame {f}: rows of the previous frame", lin_eps=None)
            want_prev[i] = o.observe()
    assert all(s.src.using_hip for s in srcs) and wfref.hip_fallback_ticks() == before
    del srcs, refs
    from helpers import ref_settings
    settings = ref_settings(cfg)
    v_hip, _ = wfref.bench("hip", settings, 64, 1, 20, 300, hop=800, seed=scenarios.SEED)
    v_avx, _ = wfref.bench("avx2", settings, 64, 1, 20, 300, hop=800, seed=scenarios.SEED)
    os.environ["WF_HIP_BATCHED_WAVE"] = "0"
    try:
        v_sync, _ = wfref.bench("hip", settings, 64, 1, 20, 300, hop=800, seed=scenarios.SEED)
    finally:
        del os.environ["WF_HIP_BATCHED_WAVE"]
    us = lambda v: 2e6 / v  # (wfref_bench counts capture channels: a stereo source = 2 per frame)
    print(f"\nplugin mode, 64 waveform displays (800 points, stereo): batched HIP {us(v_hip):.2f} us per source and frame, "
          f"synchronous HIP {us(v_sync):.2f} us, reference AVX {us(v_avx):.2f} us")
    assert wfref.hip_fallback_ticks() == before
    assert us(v_hip) < us(v_sync) / 3, "one batch per frame must be several times cheaper than 64 upload/launch/download round trips"


@pytest.mark.gpu
def test_one_waveform_source_outgrowing_the_ring_leaves_the_batch_alone():
    """Four waveform sources share a batch; one of them has an A/V-sync offset whose reserve + sweep does not fit the device ring
    (300 ms + 150 ms at 48 kHz > 16384 frames).  The C-ABI would reject the whole batch for that one stream: the host checks
    first and only that source moves to the reference's CPU class -- its rows are then the reference's of the SAME frame --
    while the other three stay in the batch, one frame late as ever."""
    import os
    from tools import synth
    wfref = _hip_env(batched=True)
    os.environ["WF_HIP_BATCH_CAPACITY"] = "64"
    cfg = scenarios.make_config(dict(waveform=1, stereo=1, width=800, meter_ms=150))
    n_src, frames, odd = 4, 50, 2
    before = wfref.hip_fallback_ticks()
    srcs = [scenarios.RefBackend(cfg, isa="hip") for _ in range(n_src)]
    refs = [scenarios.RefBackend(cfg, isa="generic") for _ in range(n_src)]
    for b in (srcs[odd], refs[odd]):
        b.set_sync_ms(300)
    assert all(s.src.using_hip for s in srcs)
    want_prev = [None] * n_src
    left_at = None
    for f in range(frames):
        for i, (s, o) in enumerate(zip(srcs, refs)):
            a = synth.block(scenarios.SEED, 700 + i, 1, 2, f * 800, 800)[0]
            for b in (s, o):
                b.push(a, muted=False)
                b.tick(1.0 / 60.0)
            got, want = s.observe(), o.observe()
            if i == odd and not s.src.using_hip:
                left_at = f if left_at is None else left_at
                if f > left_at:  # (the frame of the switch still shows what the batch had delivered)
                    assert got["silent"] == want["silent"]
                    assert np.array_equal(got["db"], want["db"]), f"source {i} frame {f}: the reference's own class, same frame"
            elif want_prev[i] is not None and i != odd:
                assert got["silent"] == want_prev[i]["silent"], f"source {i} frame {f}: m_last_silent"
                assert_db_close(got["db"], want_prev[i]["db"], f"source {i} frame {f}: rows of the previous frame", lin_eps=None)
            want_prev[i] = want
    assert left_at is not None and left_at < 30, "the reserve outgrows the ring once 300 ms of audio are behind the sync point"
    assert [s.src.using_hip for s in srcs] == [i != odd for i in range(n_src)], "only the offender leaves"
    assert wfref.hip_fallback_ticks() > before


@pytest.mark.gpu
def test_sixty_four_meter_sources_share_one_batch():
    """64 level-meter sources of one configuration: one handle, one ragged ingest + one meter_tick_kernel + one readback per
    video frame.  Every source has its own audio and amplitude, some hide, stall or lose their capture along the way; each
    one's m_meter_val at frame t+1 is what the restatement has at frame t.  Then the cost per source and frame against the
    reference's AVX tick_meter in the same harness."""
    import os
    from tools import synth
    from helpers import assert_levels_close
    wfref = _hip_env(batched=True)
    os.environ["WF_HIP_BATCH_CAPACITY"] = "64"
    cfg_dict = dict(meter=1, meter_ms=150, gravity=0.3)
    cfg = scenarios.make_config(cfg_dict)
    n_src, frames, hop = 64, 40, 800
    before = wfref.hip_fallback_ticks()
    srcs = [scenarios.RefBackend(cfg, isa="hip") for _ in range(n_src)]
    oras = [scenarios.OracleBackend(cfg) for _ in range(n_src)]
    exact = [scenarios.OracleBackend(cfg, exact=True) for _ in range(n_src)]
    assert all(s.src.using_hip for s in srcs)
    want_prev = [None] * n_src
    for f in range(frames):
        for i, (s, o, x) in enumerate(zip(srcs, oras, exact)):
            stalled = (i % 7 == 3) and f in (25, 26)         # no packet and no tick in these frames: the stream is paused
            if f == 24 and i % 5 == 1:
                s.set_hidden(True), o.set_hidden(True), x.set_hidden(True)
            if f == 28 and i % 5 == 1:
                s.set_hidden(False), o.set_hidden(False), x.set_hidden(False)
            if stalled:
                continue
            if i % 11 == 4 and f == 30:
                s.timeout(), o.timeout(), x.timeout()        # capture lost: the meter buffer is cleared, nothing is consumed
            else:
                a = synth.block(scenarios.SEED, 100 + i, 1, 2, f * hop, hop)[0] * np.float32(1.0 if i % 3 else 0.05)
                for b in (s, o, x):
                    b.push(a, muted=False)
            for b in (s, o, x):
                b.tick(1.0 / 60.0)
            got = s.observe()
            if want_prev[i] is not None:
                w, e = want_prev[i]
                assert got["silent"] == w["silent"] or got["silent"] == e["silent"], f"source {i} frame {f}"
                assert_levels_close(got["db"], w["db"], e["db"], f"source {i} frame {f}: levels of the previous frame")
            want_prev[i] = (o.observe(), x.observe())
    assert all(s.src.using_hip for s in srcs) and wfref.hip_fallback_ticks() == before
    del srcs
    settings = dict(display_mode="level_meter", rms_mode=True, meter_buf=150, temporal_smoothing="exp_moving_avg", gravity=0.3)
    v_hip, _ = wfref.bench("hip", settings, 64, 1, 20, 300, hop=hop, seed=scenarios.SEED)
    v_avx, _ = wfref.bench("avx2", settings, 64, 1, 20, 300, hop=hop, seed=scenarios.SEED)
    os.environ["WF_HIP_BATCHED_METER"] = "0"
    try:
        v_sync, _ = wfref.bench("hip", settings, 64, 1, 20, 300, hop=hop, seed=scenarios.SEED)
    finally:
        del os.environ["WF_HIP_BATCHED_METER"]
    us = lambda v: 2e6 / v  # (wfref_bench counts capture channels: a stereo source = 2 per frame)
    print(f"\nplugin mode, 64 level meters (150 ms RMS, stereo): batched HIP {us(v_hip):.2f} us per source and frame, "
          f"synchronous HIP {us(v_sync):.2f} us, reference AVX {us(v_avx):.2f} us")
    assert wfref.hip_fallback_ticks() == before
    assert us(v_hip) < us(v_sync) / 3, "one batch per frame must be several times cheaper than 64 upload/launch/download round trips"


@pytest.mark.gpu
def test_sixty_four_sources_share_one_batch():
    """64 WAVSourceHIP sources of one configuration in one fake-OBS process: one handle, one tick per video frame.  Every
    source gets its own audio, some hide or stall along the way; each one's m_decibels at frame t+1 is what the restatement
    (pinned to the reference) has at frame t.  Then the cost: microseconds per source and frame of the batched device path
    against the reference's own AVX2 class in the same harness, one host thread each."""
    import os
    from tools import synth
    wfref = _hip_env(batched=True)
    os.environ["WF_HIP_BATCH_CAPACITY"] = "64"
    cfg_dict = dict(fft_size=4096, stereo=1, slope=1.0)
    cfg = scenarios.make_config(cfg_dict)
    n_src, frames, hop = 64, 14, 800
    before = wfref.hip_fallback_ticks()
    srcs = [scenarios.RefBackend(cfg, isa="hip") for _ in range(n_src)]
    oras = [scenarios.OracleBackend(cfg) for _ in range(n_src)]
    assert all(s.src.using_hip for s in srcs)
    want_prev = [None] * n_src
    for f in range(frames):
        for i, (s, o) in enumerate(zip(srcs, oras)):
            stalled = (i % 7 == 3) and f in (5, 6)          # no packet and no tick in these frames: the stream is paused
            hidden = (i % 5 == 1) and 4 <= f < 8
            if f == 4 and i % 5 == 1:
                s.set_hidden(True), o.set_hidden(True)
            if f == 8 and i % 5 == 1:
                s.set_hidden(False), o.set_hidden(False)
            if stalled:
                continue
            a = synth.block(scenarios.SEED, 100 + i, 1, 2, f * hop, hop)[0]
            s.push(a, muted=False)
            o.push(a, muted=False)
            s.tick(1.0 / 60.0)
            o.tick(1.0 / 60.0)
            got = s.observe()
            if want_prev[i] is not None:
                assert got["silent"] == want_prev[i]["silent"], f"source {i} frame {f}"
                assert_db_close(got["db"], want_prev[i]["db"], f"source {i} frame {f}: row of the previous frame{' (hidden)' if hidden else ''}", deep=True)
            want_prev[i] = o.observe()
    assert all(s.src.using_hip for s in srcs) and wfref.hip_fallback_ticks() == before
    del srcs
    # cost per source and frame (wfref_bench: one thread ticks every source once per frame, packets through capture_audio)
    settings = dict(fft_size=4096, enable_large_fft=True, channel_mode="stereo", slope=1.0, window="hann",
                    temporal_smoothing="exp_moving_avg", gravity=0.65)
    v_hip, _ = wfref.bench("hip", settings, 64, 1, 20, 300, hop=hop, seed=scenarios.SEED)
    v_avx, _ = wfref.bench("avx2", settings, 64, 1, 20, 300, hop=hop, seed=scenarios.SEED)
    us_hip, us_avx = 2e6 / v_hip, 2e6 / v_avx  # a stereo source = 2 spectra per frame
    print(f"\nplugin mode, 64 sources x FFT 4096 stereo: batched HIP {us_hip:.1f} us per source and frame, reference AVX2 {us_avx:.1f} us")
    assert wfref.hip_fallback_ticks() == before
    assert us_hip < us_avx, f"the batched device path ({us_hip:.1f} us per source) does not beat the reference's AVX2 tick ({us_avx:.1f} us)"
    # tick + render per source and frame: the reference's AVX2 class (FMA3 interpolation in render), WAVSourceHIP with the
    # reference's render() on the host (WF_HIP_RENDER=0), and WAVSourceHIP drawing from the device's display
    for disp, extra in (("26 Lanczos bars", dict(display_mode="bars", interp_mode="lanczos")), ("800-point Catmull-Rom curve", dict(display_mode="curve", interp_mode="catmull_rom"))):
        st = dict(settings, **extra)
        drawn, on_host = wfref.hip_device_renders(), wfref.hip_host_renders()
        v_dev, _ = wfref.bench("hip", st, 64, 1, 20, 300, hop=hop, seed=scenarios.SEED, render=True)
        assert wfref.hip_device_renders() > drawn and wfref.hip_host_renders() - on_host <= 64, "the timed frames were not drawn from the device's display"
        os.environ["WF_HIP_RENDER"] = "0"
        try:
            v_host, _ = wfref.bench("hip", st, 64, 1, 20, 300, hop=hop, seed=scenarios.SEED, render=True)
        finally:
            del os.environ["WF_HIP_RENDER"]
        v_ref, _ = wfref.bench("avx2", st, 64, 1, 20, 300, hop=hop, seed=scenarios.SEED, render=True)
        print(f"plugin mode, 64 sources x FFT 4096 stereo, {disp}, tick + render per source and frame: display from the device "
              f"{2e6 / v_dev:.1f} us, device rows + the reference's render on the host {2e6 / v_host:.1f} us, reference AVX2 {2e6 / v_ref:.1f} us")
        assert v_dev > v_ref


@pytest.mark.gpu
def test_canary_guards_behind_every_device_block(monkeypatch):
    """there is no compute-sanitizer on this stack, so with WF_HIP_CANARY=1 every device block of a handle
    ends in 256 guard bytes that wf_hip_sync reads back.  Every golden scenario once more with the guards armed -- spectrum at
    every kind of FFT size, bars, curves, filters, vertex fill, level meter, waveform display, volume normalisation -- and a sync
    after every tick: no kernel writes past a buffer.  And the check itself: one word written behind m_decibels (through the
    HIP runtime, from outside the library) turns the next sync into WF_HIP_ERR_RUNTIME naming the block."""
    import ctypes as C
    import waveform_amd as wf
    monkeypatch.setenv("WF_HIP_CANARY", "1")
    for name in NAMES:
        sc = scenarios.SCENARIOS[name]
        cfg = scenarios.make_config(sc["cfg"])
        be = scenarios.HipBackend(cfg, streams=5, probe=2)
        try:
            real_tick = be.tick

            def tick(seconds, _t=real_tick, _b=be):
                _t(seconds)
                _b.batch.sync()  # raises WfHipError if a guard was touched
            be.tick = tick
            scenarios.play(be, sc)
            be.batch.sync()
        finally:
            be.close()
    cfg = wf.Config.defaults(fft_size=2048, stereo=1, slope=1.0, bars=1, interp_mode=wf.INTERP["lanczos"])
    with wf.SpectrumBatch(cfg, 3) as b:
        b.push_synth(scenarios.SEED, 0, 800)
        b.tick()
        b.sync()
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        end_of_rows = b.decibels_device_ptr() + 3 * b.output_channels * b.bins * 4
        assert hip.hipMemset(C.c_void_p(end_of_rows + 8), 0, 4) == 0 and hip.hipDeviceSynchronize() == 0
        with pytest.raises(wf.WfHipError) as e:
            b.sync()
        assert "WF_HIP_CANARY" in str(e.value) and "past its end" in str(e.value), str(e.value)
