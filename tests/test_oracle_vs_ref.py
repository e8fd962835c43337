"""The CPU restatement (oracle/wf_oracle*.c) against the reference itself (oracle/_ref/libwfref.so: phandasm/waveform's own
translation units + vendored FFTW), live, on the random cases of tests/test_gpu_fuzz.py -- the same draws the GPU suite
plays against the restatement.  Pins the oracle beyond the hand-written golden scenarios: configurations and event scripts
nobody chose, including the reference's own get_settings() repairs and its own update_input_rms().  No GPU involved.
"""
import pytest

import scenarios
import test_gpu_fuzz as fuzz
from oracle import wfref

STEP = 4  # every 4th seed of each family (the GPU suite plays every 8th against libwfref as well)


def _play_pair(cfg_dict, steps, what, sync_ms=0):
    if not wfref.available():
        pytest.skip("oracle/_ref/libwfref.so not built")
    cfg = scenarios.make_config(cfg_dict)
    sc = dict(cfg=cfg_dict, steps=steps, record="all", sync_ms=sync_ms)
    got = scenarios.play(scenarios.OracleBackend(cfg), sc)
    want = scenarios.play(scenarios.RefBackend(cfg), sc)
    return cfg, got, want


@pytest.mark.parametrize("seed", range(0, len(fuzz.SPEC_SEEDS), STEP))
def test_restatement_matches_reference_pow2(seed):
    cfg_dict, steps, sync_ms = fuzz.draw(seed, "pow2")
    cfg, got, want = _play_pair(cfg_dict, steps, f"pow2 {seed}", sync_ms=sync_ms)
    fuzz._compare(got, want, fuzz._undo_db(cfg), f"pow2 case {seed} ({cfg_dict}): restatement vs libwfref", cfg_stepped=cfg_dict.get("vertices") == 3, cfg=cfg)


@pytest.mark.parametrize("seed", range(0, len(fuzz.BLU_SEEDS), STEP))
def test_restatement_matches_reference_any_size(seed):
    cfg_dict, steps, sync_ms = fuzz.draw(seed, "any")
    cfg, got, want = _play_pair(cfg_dict, steps, f"any {seed}", sync_ms=sync_ms)
    fuzz._compare(got, want, fuzz._undo_db(cfg), f"any-size case {seed} ({cfg_dict}): restatement vs libwfref", cfg_stepped=cfg_dict.get("vertices") == 3, cfg=cfg)


def _huge_seeds(count=10):
    """the first seeds of the huge family whose length the restatement's O(n p) DFT can do in reasonable time (the GPU suite
    checks the others against libwfref.so alone)"""
    out = []
    for s in fuzz.HUGE_SEEDS:
        if fuzz._largest_prime_factor(fuzz.draw(s, "huge")[0]["fft_size"]) <= fuzz.RESTATEMENT_MAX_PRIME:
            out.append(s)
    return out[::max(1, len(out) // count)][:count]


@pytest.mark.parametrize("seed", _huge_seeds())
def test_restatement_matches_reference_huge_size(seed):
    cfg_dict, steps, sync_ms = fuzz.draw(seed, "huge")
    cfg, got, want = _play_pair(cfg_dict, steps, f"huge {seed}", sync_ms=sync_ms)
    fuzz._compare(got, want, fuzz._undo_db(cfg), f"huge-size case {seed} ({cfg_dict}): restatement vs libwfref", cfg_stepped=cfg_dict.get("vertices") == 3, cfg=cfg)


@pytest.mark.parametrize("seed", range(0, len(fuzz.WIDE_SEEDS), STEP))
def test_restatement_matches_reference_wide_ranges(seed):
    """the reference's full slider ranges (fuzz.draw_wide)"""
    cfg_dict, steps, sync_ms = fuzz.draw_wide(seed)
    cfg, got, want = _play_pair(cfg_dict, steps, f"wide {seed}", sync_ms=sync_ms)
    fuzz._compare(got, want, fuzz._undo_db(cfg), f"wide-range case {seed} ({cfg_dict}): restatement vs libwfref", cfg_stepped=cfg_dict.get("vertices") == 3, cfg=cfg)


@pytest.mark.parametrize("seed", range(0, len(fuzz.METER_SEEDS), STEP))
def test_restatement_matches_reference_meter(seed):
    import numpy as np
    from helpers import assert_db_close
    cfg_dict, steps = fuzz.draw_meter(seed)
    cfg, got, want = _play_pair(cfg_dict, steps, f"meter {seed}")
    for t, (g, w) in enumerate(zip(got, want)):
        assert g["silent"] == w["silent"], f"meter case {seed} tick {t}"
        # the restatement adds in the reference's order: the levels agree to the last bits of log10f
        assert_db_close(g["db"], w["db"], f"meter case {seed} tick {t} levels ({cfg_dict})", lin_eps=None)
        assert np.all(np.abs(g["bars"].astype(np.float64) - w["bars"]) <= 1e-5 * np.abs(w["bars"]) + 2e-3), f"meter case {seed} tick {t} bars"


@pytest.mark.parametrize("seed", range(0, len(fuzz.WAVE_SEEDS), 2))
def test_restatement_matches_reference_waveform(seed):
    from helpers import assert_db_close
    cfg_dict, steps, sync_ms = fuzz.draw_wave(seed)
    cfg, got, want = _play_pair(cfg_dict, steps, f"wave {seed}", sync_ms=sync_ms)
    for t, (g, w) in enumerate(zip(got, want)):
        assert g["silent"] == w["silent"], f"wave case {seed} tick {t}"
        assert_db_close(g["db"], w["db"], f"wave case {seed} tick {t} rows ({cfg_dict}, sync {sync_ms} ms)", lin_eps=None)
