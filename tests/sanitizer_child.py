"""TEST HARNESS child process of tests/test_sanitizers.py: runs with a sanitizer runtime preloaded and WFREF_LIBRARY pointing at the
matching build of the reference harness (oracle/_ref/libwfref_asan.so / libwfref_tsan.so).
usage: python sanitizer_child.py scenarios|threads ISA [seconds]"""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import numpy as np  # noqa: E402
from oracle import wfref  # noqa: E402
import scenarios  # noqa: E402

what, isa = sys.argv[1], sys.argv[2]
if isa == "hip":
    os.environ.setdefault("WF_HIP_LIBRARY", str(ROOT / "waveform_amd" / "libwaveform_hip.so"))
if what == "scenarios":
    # golden scenarios through the reference's own update / capture_audio / tick / render with the harness, the fake libobs and
    # (isa hip) the binding instrumented: one of every display kind, the silence and hide / show state machines, ragged hops
    names = ["default_4096_stereo_ema_slope", "silence_cycle", "hide_show", "ragged_hops", "verts_bars_4096_stereo_caps",
             "verts_curve_1024_line", "verts_stepped_2048_stereo", "meter_rms_stereo", "normalize_default"]
    names = [n for n in names if n in scenarios.SCENARIOS] or sorted(scenarios.SCENARIOS)[:10]
    for name in names:
        sc = scenarios.SCENARIOS[name]
        be = scenarios.RefBackend(scenarios.make_config(sc["cfg"]), isa=isa)
        recs = scenarios.play(be, sc)
        assert recs and all(np.all(np.isfinite(r["db"])) for r in recs), name
    print("scenarios ok", len(names))
else:
    seconds = float(sys.argv[3]) if len(sys.argv) > 3 else 1.5
    st = dict(fft_size=2048, channel_mode="stereo", slope=1.0, display_mode="bars", interp_mode="lanczos")
    differ, stats = wfref.thread_stress(isa, st, 8 if isa != "hip" else 24, seconds)
    assert differ == 0, f"{differ} sources differ from a fresh one after the run: {stats}"
    assert stats["ticks"] > 50 and stats["packets"] > 50 and stats["updates"] > 10, stats
    if isa == "hip":
        assert wfref.hip_fallback_ticks() == 0, "ticks fell back to the CPU class"
    print("threads ok", stats)
