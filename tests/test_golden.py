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
