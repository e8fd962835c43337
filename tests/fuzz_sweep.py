"""one-off extended fuzz sweep (development aid): python tests/fuzz_sweep.py LO HI [spec,meter,wave]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, traceback
import scenarios, test_gpu_fuzz as f
import waveform_amd as wf
from helpers import assert_db_close
bad = 0
def run(kind, seed):
    global bad
    try:
        if kind == "spec":
            cfg_dict, steps = f.draw(seed); sync = 0
        elif kind == "meter":
            cfg_dict, steps = f.draw_meter(seed); sync = 0
        else:
            cfg_dict, steps, sync = f.draw_wave(seed)
        cfg = scenarios.make_config(cfg_dict)
        sc = dict(cfg=cfg_dict, steps=steps, record="all", sync_ms=sync)
        rms = 0.0316 if (cfg.normalize_volume and kind != "meter") else None
        try:
            hip = scenarios.HipBackend(cfg, streams=2, probe=1, input_rms=rms)
        except wf.WfHipError as e:
            assert e.code == -2; return
        ora = scenarios.OracleBackend(cfg, input_rms=rms)
        try:
            got = scenarios.play(hip, sc); want = scenarios.play(ora, sc)
        finally:
            hip.close()
        for t, (g, w) in enumerate(zip(got, want)):
            assert g["silent"] == w["silent"], f"silent tick {t}"
            assert_db_close(g["db"], w["db"], f"tick {t}")
            if w["bars"] is not None:
                err = np.abs(g["bars"].astype(np.float64) - w["bars"])
                assert np.all(err <= 1e-5 * np.abs(w["bars"]) + 2e-3), f"bars tick {t} {err.max()}"
    except Exception as e:
        bad += 1
        print("FAIL", kind, seed, str(e)[:300], flush=True)
lo, hi = int(sys.argv[1]), int(sys.argv[2])
kinds = sys.argv[3].split(",") if len(sys.argv) > 3 else ["spec", "meter", "wave"]
for s in range(lo, hi):
    for k in kinds:
        run(k, s)
print("done", lo, hi, "failures", bad)
