"""extended fuzz sweep (development aid): python tests/fuzz_sweep.py LO HI [pow2,any,huge,meter,wave,dropin-pow2,dropin-any,dropin-meter,dropin-wave]
runs the same case functions as tests/test_gpu_fuzz.py over seeds [LO, HI) and lists failures and skips"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pytest  # noqa: E402
import test_gpu_fuzz as f  # noqa: E402

lo, hi = int(sys.argv[1]), int(sys.argv[2])
kinds = sys.argv[3].split(",") if len(sys.argv) > 3 else ["pow2", "any", "meter", "wave"]
run = {"pow2": lambda s: f.run_spectrum_case(s, "pow2"), "any": lambda s: f.run_spectrum_case(s, "any"),
       "huge": lambda s: f.run_spectrum_case(s, "huge"), "meter": f.run_meter_case,
       "wave": f.test_hip_waveform_matches_oracle_on_random_case}


def run_dropin_case(seed, family):
    """the fuzz script of `family` through the reference plugin itself, once with its own CPU class and once with WAVSourceHIP
    (synchronous mode) as the tick implementation; the device path must stay in use, no tick may fall back"""
    import numpy as np
    from pathlib import Path
    import scenarios
    from oracle import wfref
    from helpers import assert_db_close
    os.environ["WF_HIP_LIBRARY"] = str(Path(ROOT) / "waveform_amd" / "libwaveform_hip.so")
    os.environ["WF_HIP_BATCHED"] = "0"
    if family == "meter":
        cfg_dict, steps = f.draw_meter(seed)
        sync_ms = 0
    elif family == "wave":
        cfg_dict, steps, sync_ms = f.draw_wave(seed)
    else:
        cfg_dict, steps, sync_ms = f.draw(seed, family)
    cfg_dict = dict(cfg_dict)
    cfg_dict.pop("vertices", None)  # the plugin's render loop stays the reference's own
    cfg = scenarios.make_config(cfg_dict)
    sc = dict(cfg=cfg_dict, steps=steps, record="all", sync_ms=sync_ms)
    before = wfref.hip_fallback_ticks()
    hip = scenarios.RefBackend(cfg, isa="hip")
    assert hip.src.using_hip, "WAVSourceHIP did not take the device path"
    got = scenarios.play(hip, sc)
    assert hip.src.using_hip and wfref.hip_fallback_ticks() == before, "fell back to the CPU class"
    want = scenarios.play(scenarios.RefBackend(cfg, isa="generic"), sc)
    assert len(got) == len(want)
    undo = f._undo_db(cfg) if family in ("pow2", "any") else None
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
        if False:
            pass
        else:
            assert_db_close(g["db"], w["db"], what + " rows", undo_db=undo, **({} if family in ("pow2", "any") else {"lin_eps": None}))


def run_dropin_batched_case(seed, family):
    """spectrum scripts through the plugin's batched mode (sources share a handle, rows read one frame late)"""
    import numpy as np
    from pathlib import Path
    import scenarios
    import test_golden as tg
    from oracle import wfref
    from helpers import assert_db_close
    os.environ["WF_HIP_LIBRARY"] = str(Path(ROOT) / "waveform_amd" / "libwaveform_hip.so")
    os.environ["WF_HIP_BATCHED"] = "1"
    cfg_dict, steps, sync_ms = f.draw(seed, family)
    cfg_dict = dict(cfg_dict)
    cfg_dict.pop("vertices", None)
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
    undo = f._undo_db(cfg)
    for t, (g, w) in enumerate(zip(got, want)):
        what = f"batched drop-in {family} case {seed} tick {t} ({cfg_dict}, sync {sync_ms} ms)"
        assert g["silent"] == w["silent"], what + f": m_last_silent {g['silent']} != {w['silent']}"
        assert_db_close(g["db"], w["db"], what + " rows", undo_db=undo)


for _fam in ("pow2", "any"):
    run["batched-" + _fam] = (lambda fam: (lambda s: run_dropin_batched_case(s, fam)))(_fam)
for _fam in ("pow2", "any", "meter", "wave"):
    run["dropin-" + _fam] = (lambda fam: (lambda s: run_dropin_case(s, fam)))(_fam)

bad = skipped = 0
for k in kinds:
    for s in range(lo, hi):
        try:
            run[k](s)
        except pytest.skip.Exception as e:
            skipped += 1
            print("SKIP", k, s, str(e)[:200], flush=True)
        except Exception as e:
            bad += 1
            m = str(e).replace("\n", " ")
            print("FAIL", k, s, m[-260:] if len(m) > 260 else m, flush=True)
            if os.environ.get("WF_FUZZ_VERBOSE"):
                print("     ", m[:700], flush=True)
print("done", lo, hi, kinds, "failures", bad, "skips", skipped)
