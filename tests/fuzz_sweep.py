"""extended fuzz sweep (development aid): python tests/fuzz_sweep.py LO HI [pow2,any,huge,meter,wave,dropin-pow2,dropin-any,dropin-meter,dropin-wave,batched-pow2,batched-any,batched-huge,batched-meter,batched-wave]
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
       "huge": lambda s: f.run_spectrum_case(s, "huge"), "smooth": lambda s: f.run_spectrum_case(s, "smooth"), "wide": lambda s: f.run_spectrum_case(s, "wide"), "meter": f.run_meter_case, "meter-wide": lambda s: f.run_meter_case(s, wide=True),
       "wave": f.test_hip_waveform_matches_oracle_on_random_case}


for _fam in ("pow2", "any", "huge"):
    run["batched-" + _fam] = (lambda fam: (lambda s: f.run_dropin_batched_case(s, fam)))(_fam)
run["batched-meter"] = f.run_dropin_batched_meter_case
run["batched-wave"] = f.run_dropin_batched_wave_case
for _fam in ("pow2", "any", "huge", "meter", "wave"):
    run["dropin-" + _fam] = (lambda fam: (lambda s: f.run_dropin_case(s, fam)))(_fam)

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
print("display checks", f.ARM["display_checks"], "second arm", f.ARM["display_arm"], f.ARM["arm_cases"][:20])
for u in f.ARM["unsupported"]:
    print("UNSUPPORTED", u[-400:])
print("done", lo, hi, kinds, "failures", bad, "skips", skipped, "unsupported", len(f.ARM["unsupported"]))
