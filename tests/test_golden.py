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
        assert_db_close(r["db"], z[f"db_{t}"], f"{name} tick {t} decibels")
        if f"bars_{t}" in z.files:
            got, want = r["bars"], z[f"bars_{t}"]
            assert got is not None
            err = np.abs(got.astype(np.float64) - want)
            assert np.all(err <= 1e-5 * np.abs(want) + 2e-3), f"{name} tick {t} bars: max err {err.max():.3e} px"


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


# ---- the drop-in: the reference plugin itself, with only tick_spectrum replaced by the HIP binding -----------------
DROPIN = ["cfg1_mono_1024", "cfg2_stereo_2048_nosmooth", "cfg3_stereo_4096_ema_slope", "cfg4_16384_tv_lanczos_bars",
          "mono_mix_4096_tv_fastpeaks", "rolloff_catrom_linear", "ragged_hops", "silence_cycle", "half_silent_stereo",
          "hide_show", "muted_packets", "timeout_spectrum", "split_8192_half_silent", "small_512_stereo_bars", "small_128_single_dup_curve", "large_32768_single_tv", "any_800_mono_mix_bars", "any_4160_stereo_silence",
          # WAVSourceHIP::tick_meter
          "meter_rms_stereo", "meter_peak_mono_tv_fastpeaks", "meter_nosmooth_ragged", "meter_silence_cycle", "meter_half_silent",
          "meter_hide_show_timeout",
          # WAVSourceHIP::tick_waveform
          "wave_stereo_800", "wave_mono_mix_ragged", "wave_single_dup_stall", "wave_hide_timeout_sync", "wave_normalize"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", DROPIN)
def test_reference_plugin_with_hip_tick(name):
    """oracle/_ref's WAVSource (update / capture_audio / tick / render_bars run verbatim) with WAVSourceHIP
    (host/wav_source_hip.cpp) as the tick_spectrum / tick_meter implementation, against the same golden vectors."""
    import os
    from oracle import wfref
    if not wfref.available():
        pytest.skip("oracle/_ref/libwfref.so not built")
    os.environ["WF_HIP_LIBRARY"] = str(Path(__file__).resolve().parent.parent / "waveform_amd" / "libwaveform_hip.so")
    cfg = scenarios.make_config(scenarios.SCENARIOS[name]["cfg"])
    be = scenarios.RefBackend(cfg, isa="hip")
    assert be.src.using_hip, "WAVSourceHIP fell back to the CPU path: the HIP library did not load or no gfx950 device"
    _check(name, be)
