"""the reference ships without sanitizers; its concurrency contract is one recursive timed mutex per source
(src/source.hpp:98-101) with a 10 ms try-lock on the audio thread (src/source.cpp:1822-1824).  Here the host-side code of this
repository runs under AddressSanitizer + UndefinedBehaviorSanitizer and ThreadSanitizer:

  * waveform_amd/csrc/wf_hip_multi.cpp (the multi-device group: worker threads, shard arithmetic, peer gather, failure
    handling) over a host-only mock of the device layer (tests/mock/) -- CPU suite;
  * the reference harness with the fake libobs and host/wav_source_hip.cpp (oracle/_ref/libwfref_asan.so / _tsan.so): golden
    scenarios, and the source under real threads (audio thread per source, video thread, UI thread calling update / show / hide /
    destroy + create) -- the reference's own classes in the CPU suite, WAVSourceHIP in batched mode on the GPU box;
  * wf_host_tables.cpp + the wavefront emulator's host code (build/libwfemu_asan.so): every table builder over fuzzed
    configurations.

Every run is a child process with the sanitizer's runtime preloaded; a report fails the test (non-zero exit / text on stderr)."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _runtime(name):
    p = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    if not p or not Path(p).is_absolute() or not Path(p).exists():
        pytest.skip(f"{name} not installed")
    return p


def _child(args, preload, env_extra=None, timeout=600):
    env = dict(os.environ, LD_PRELOAD=preload, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", TSAN_OPTIONS=f"halt_on_error=1:second_deadlock_stack=1:suppressions={ROOT / 'tests' / 'tsan.supp'}",
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.update(env_extra or {})
    r = subprocess.run([sys.executable] + [str(a) for a in args], capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    report = [l for l in r.stderr.splitlines() if "Sanitizer" in l or "runtime error" in l]
    assert r.returncode == 0 and not report, (r.stdout[-1500:], r.stderr[-4000:])
    return r.stdout


def _built(path, make_dir):
    if not path.exists():
        subprocess.run(["make", "-C", str(make_dir)], capture_output=True)
    if not path.exists():
        pytest.skip(f"{path} not built")
    return path


@pytest.mark.parametrize("san", ["asan", "tsan"])
def test_multi_device_group_over_the_mock_device(san):
    lib = _built(ROOT / "build" / f"libwfmulti_{san}.so", ROOT / "tests" / "mock")
    out = _child([ROOT / "tests" / "mock" / "multi_stress.py", lib], _runtime("libasan.so" if san == "asan" else "libtsan.so"))
    assert "multi stress ok" in out


@pytest.mark.parametrize("isa", ["generic", "avx2"])
def test_reference_harness_scenarios_under_asan_ubsan(isa):
    lib = ROOT / "oracle" / "_ref" / "libwfref_asan.so"
    if not lib.exists():
        pytest.skip("oracle/_ref/libwfref_asan.so not built (make -C oracle/ref where /root/reference exists)")
    out = _child([ROOT / "tests" / "sanitizer_child.py", "scenarios", isa], _runtime("libasan.so"), {"WFREF_LIBRARY": str(lib)})
    assert "scenarios ok" in out


@pytest.mark.parametrize("san", ["asan", "tsan"])
def test_source_under_obs_threads_reference_classes(san):
    """audio thread per source, video thread, UI thread (update with changing FFT sizes, show / hide, destroy + create) on the
    reference's AVX2 class: the harness, the fake libobs' audio_cb_mutex and the reference's m_mtx discipline hold under
    ThreadSanitizer / AddressSanitizer; afterwards every source computes what a fresh one computes"""
    lib = ROOT / "oracle" / "_ref" / f"libwfref_{san}.so"
    if not lib.exists():
        pytest.skip(f"{lib} not built")
    out = _child([ROOT / "tests" / "sanitizer_child.py", "threads", "avx2", "1.5"], _runtime("libasan.so" if san == "asan" else "libtsan.so"),
                 {"WFREF_LIBRARY": str(lib)})
    assert "threads ok" in out


def test_host_tables_under_asan_ubsan():
    lib = _built(ROOT / "build" / "libwfemu_asan.so", ROOT / "tests" / "emu")
    code = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import os
os.environ["WFEMU_LIBRARY"] = %r
import numpy as np
import emu_binding as emu, scenarios
import test_gpu_fuzz as fz
n = 0
for fam, seeds in (("pow2", range(0, 60)), ("any", range(0, 40)), ("huge", range(0, 12)), ("smooth", range(0, 20))):
    for s in seeds:
        cfg_dict, _, _ = fz.draw(s, fam)
        cfg = scenarios.make_config(cfg_dict)
        for k in (0, 1, 2, 3, 5, 7, 8, 9):
            emu.host_table(cfg, k)
        if cfg.bars and cfg.fft_size >= 512 and (cfg.fft_size & (cfg.fft_size - 1)) == 0 and cfg.fft_size <= 32768:
            T, P = {512: (64, 4), 1024: (64, 8), 2048: (64, 16), 4096: (128, 16), 8192: (256, 16), 16384: (512, 16), 32768: (512, 32)}[int(cfg.fft_size)]
            emu.bar_pieces(cfg, T, P, P // 4 + 2, 0); emu.bar_lanes(cfg, T, P // 4 + 2, 0)
        n += 1
for s in range(60):  # the reference's full slider ranges (some combinations have no table of that kind: an error code, not a report)
    cfg_dict, _, _ = fz.draw_wide(s)
    for k in (0, 1, 2, 3, 5, 7, 8, 9):
        try:
            emu.host_table(scenarios.make_config(cfg_dict), k)
        except ValueError:
            pass
# the emulated tick of every geometry under the same build: the spectrum's exchange buffer is a heap block of exactly Geom::LDS_CF
# complex points, so an LDS index past it is a report here (and a -77 from the emulator's own bounds check anywhere)
from tools import synth
for nfft in (512, 1024, 2048, 4096, 8192, 16384, 32768):
    cfg = scenarios.make_config(dict(fft_size=nfft, stereo=1, slope=1.0))
    ring = np.ascontiguousarray(synth.block(1, 0, 1, 2, 0, 2 * nfft)[0], np.float32)
    emu.tick(cfg, ring, 2 * nfft, np.zeros((2, nfft // 2), np.float32))
# the tables of the rows-by-Bluestein form of the sizes above 16384 (wf::build_bluestein_rows): the row counts and lengths the plan picks
for nfft, c in ((16400, 8), (17728, 16), (32704, 16), (33472, 16), (48016, 8), (48064, 16), (65344, 16), (65424, 8), (65472, 16)):
    L, rowtw, bhat, q = emu.bluestein_rows(nfft // 2, c)
    assert L >= 2 * (nfft // 2 // c) - 1 and rowtw.shape == (c, nfft // 2 // c) and bhat.size == L and q.size == nfft // 2 // c
print("tables ok", n)
""" % (str(ROOT), str(ROOT / "tests"), str(lib))
    out = _child(["-c", code], _runtime("libasan.so"))
    assert "tables ok" in out


def test_source_under_obs_threads_wavsourcehip_over_a_mock_library_tsan():
    """ThreadSanitizer and the HIP runtime do not share a process (the runtime dies while it loads, before any of this code
    runs), so the binding's own synchronisation -- the process-wide registry mutex against every source's m_mtx, members joining
    and leaving groups while a frame is being assembled, the double-buffered staging blocks -- is checked here against a
    host-only libwaveform_hip stand-in (tests/mock/mock_wf_hip.cpp) that host/wav_source_hip.cpp dlopen()s like the real one:
    24 sources in batched mode, audio thread each, video thread, UI thread; no tick may fall back to the CPU class."""
    lib = ROOT / "oracle" / "_ref" / "libwfref_tsan.so"
    if not lib.exists():
        pytest.skip(f"{lib} not built")
    mock = _built(ROOT / "build" / "libwfhip_mock.so", ROOT / "tests" / "mock")
    out = _child([ROOT / "tests" / "sanitizer_child.py", "threads", "hip", "4"], _runtime("libtsan.so"),
                 {"WFREF_LIBRARY": str(lib), "WF_HIP_BATCHED": "1", "WF_HIP_LIBRARY": str(mock)})
    assert "threads ok" in out


@pytest.mark.gpu
@pytest.mark.parametrize("san", ["asan"])
def test_source_under_obs_threads_wavsourcehip_batched(san):
    """The same threads on WAVSourceHIP in batched mode: 24 sources share device groups (the registry mutex against every
    source's m_mtx; members joining and leaving groups as update() changes their FFT size; destroy + create under a flush), with
    host/wav_source_hip.cpp instrumented -- ThreadSanitizer for the lock order and the shared group state, AddressSanitizer
    for the staging buffers.  No tick may fall back to the CPU class; afterwards every source computes what a fresh one does."""
    lib = ROOT / "oracle" / "_ref" / f"libwfref_{san}.so"
    if not lib.exists():
        pytest.skip(f"{lib} not built")
    out = _child([ROOT / "tests" / "sanitizer_child.py", "threads", "hip", "10"],
                 _runtime("libasan.so" if san == "asan" else "libtsan.so"), {"WFREF_LIBRARY": str(lib), "WF_HIP_BATCHED": "1"}, timeout=900)
    assert "threads ok" in out


@pytest.mark.gpu
def test_wavsourcehip_scenarios_under_asan_ubsan():
    lib = ROOT / "oracle" / "_ref" / "libwfref_asan.so"
    if not lib.exists():
        pytest.skip("oracle/_ref/libwfref_asan.so not built")
    for batched in ("0", "1"):
        out = _child([ROOT / "tests" / "sanitizer_child.py", "scenarios", "hip"], _runtime("libasan.so"), {"WFREF_LIBRARY": str(lib), "WF_HIP_BATCHED": batched})
        assert "scenarios ok" in out
