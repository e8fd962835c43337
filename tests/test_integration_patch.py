"""The integration applied, not described.  integration/waveform-hip.patch is what a maintainer applies to phandasm/waveform
(the factory line in callbacks::create, src/source.cpp:87-102, and the CMake source list, CMakeLists.txt:128-183);
oracle/ref/Makefile applies it with patch(1) to a scratch copy, compiles the patched source.cpp in place of the original and
links oracle/_ref/libwfref_plugin.so -- the plugin with the patch in.  Here sources are created through the reference's OWN
registered obs_source_info::create (wfref_create(isa="create")), not by the harness picking a class:
  * without a GPU the patched factory must fall through to the reference's AVX2 / AVX / generic class and reproduce the goldens;
  * on the GPU box it must instantiate WAVSourceHIP, keep the device path for whole scenarios, and reproduce the goldens in
    both plugin modes (synchronous, and batched one frame late).
Each check runs in a child process: the library under test is chosen by WFREF_LIBRARY when oracle.wfref is first imported."""
import os
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
PLUGIN = ROOT / "oracle" / "_ref" / "libwfref_plugin.so"
PATCH = ROOT / "integration" / "waveform-hip.patch"
REF = Path(os.environ.get("WF_REFERENCE", "/root/reference"))


def _child(code, batched=None, timeout=900):
    env = dict(os.environ, WFREF_LIBRARY=str(PLUGIN), WF_HIP_LIBRARY=str(ROOT / "waveform_amd" / "libwaveform_hip.so"))
    if batched is not None:
        env["WF_HIP_BATCHED"] = "1" if batched else "0"
    pre = f"import sys\nsys.path[:0] = [{str(ROOT)!r}, {str(ROOT / 'tests')!r}]\n"
    return subprocess.run([sys.executable, "-c", pre + code], capture_output=True, text=True, timeout=timeout, env=env)


def test_patch_applies_cleanly_to_the_reference_tree(tmp_path):
    """patch(1) takes integration/waveform-hip.patch against the reference's two files without fuzz or rejects, and the result
    has WAVSourceHIP first in callbacks::create and the binding in PLUGIN_SOURCES"""
    if not (REF / "src" / "source.cpp").exists():
        pytest.skip("the reference tree is not on this box (the prebuilt oracle/_ref/libwfref_plugin.so travels instead)")
    (tmp_path / "src").mkdir()
    shutil.copy(REF / "src" / "source.cpp", tmp_path / "src" / "source.cpp")
    shutil.copy(REF / "CMakeLists.txt", tmp_path / "CMakeLists.txt")
    r = subprocess.run(["patch", "-p1", "--no-backup-if-mismatch", "-d", str(tmp_path)], stdin=open(PATCH), capture_output=True, text=True)
    assert r.returncode == 0 and "fuzz" not in r.stdout and "FAILED" not in r.stdout, (r.stdout, r.stderr)
    src = (tmp_path / "src" / "source.cpp").read_text()
    create = src[src.index("static void *create("):src.index("static void destroy(")]
    assert create.index("WAVSourceHIP::available()") < create.index("WAVSource::HAVE_AVX2")
    assert create.count("new WAVSourceHIP(source)") == 2  # with and without ENABLE_X86_SIMD
    cm = (tmp_path / "CMakeLists.txt").read_text()
    assert '"src/wav_source_hip.cpp"' in cm and "${CMAKE_DL_LIBS}" in cm


CPU_CODE = r'''
import numpy as np, scenarios, test_golden as tg
from oracle import wfref
assert wfref.LIB_PATH.name == "libwfref_plugin.so"
for name in ("cfg3_stereo_4096_ema_slope", "cfg5_4096_bars", "silence_cycle", "meter_rms_stereo"):
    cfg = scenarios.make_config(scenarios.SCENARIOS[name]["cfg"])
    be = scenarios.RefBackend(cfg, isa="create")       # obs_source_info::create of the patched plugin
    cls = be.src.class_name
    assert not be.src.using_hip and cls in ("avx2", "avx", "generic"), cls
    tg._check(name, be)                                 # the reference's own SIMD classes stay inside the golden tolerance
print("cpu factory ok", cls)
'''


def test_patched_factory_falls_back_to_the_cpu_classes_without_a_device():
    import waveform_amd as wf
    if wf.device_count() > 0:
        pytest.skip("a GPU is present: the patched factory takes WAVSourceHIP (test_patched_factory_creates_wavsourcehip)")
    if not PLUGIN.exists():
        pytest.skip("oracle/_ref/libwfref_plugin.so not built")
    r = _child(CPU_CODE)
    assert r.returncode == 0 and "cpu factory ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


GPU_CODE = r'''
import os, numpy as np, scenarios, test_golden as tg
from oracle import wfref
assert wfref.LIB_PATH.name == "libwfref_plugin.so"
batched = os.environ["WF_HIP_BATCHED"] == "1"
names = tg.SPECTRUM_DROPIN if batched else tg.DROPIN
before, rms_before = wfref.hip_fallback_ticks(), wfref.hip_host_rms_updates()
for name in names:
    sc = scenarios.SCENARIOS[name]
    cfg = scenarios.make_config(sc["cfg"])
    be = scenarios.RefBackend(cfg, isa="create")       # the patched callbacks::create picks the class
    assert be.src.class_name == "hip" and be.src.using_hip, (name, be.src.class_name)
    if not batched:
        tg._check(name, be)
    else:
        z, meta = tg._load(name)
        late = tg._OneFrameLate(be)
        scenarios.play(late, sc)
        recs = late.finish()
        assert len(recs) == meta["n_ticks"]
        assert np.array_equal(np.array([r["silent"] for r in recs], np.uint8), z["silent"]), name
        for t, r in scenarios.recorded(recs, sc["record"]):
            tg.assert_db_close(r["db"], z[f"db_{t}"], f"{name} tick {t} decibels, one frame late, through callbacks::create", deep=True)
    assert be.src.using_hip, name
assert wfref.hip_fallback_ticks() == before, "ticks were served by the CPU class"
if batched:
    assert wfref.hip_host_rms_updates() == rms_before
print("gpu factory ok", len(names), "scenarios", "batched" if batched else "synchronous")
'''


@pytest.mark.gpu
@pytest.mark.parametrize("batched", [False, True])
def test_patched_factory_creates_wavsourcehip(batched):
    """every drop-in golden scenario with the source coming out of the patched plugin's own callbacks::create"""
    if not PLUGIN.exists():
        pytest.skip("oracle/_ref/libwfref_plugin.so not built")
    r = _child(GPU_CODE, batched=batched)
    assert r.returncode == 0 and "gpu factory ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
