"""The single-process multi-device group of the C ABI (wf_hip_multi_*, include/wf_hip.h): shards of
one batch on their own host threads, results identical to one plain handle, and the all-gather of the bar heights over every
transport this box can run -- "local" (one shard), "peer" (several shards; the same device may be named more than once, so a
1-GPU box exercises the shard arithmetic, the threads, the cross-stream ordering and the double buffering), "rccl" (ncclAllGather
of the dlopen()ed librccl.so: one rank on a 1-GPU box, every visible device where there are more)."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import waveform_amd as wf
from tools import synth

pytestmark = pytest.mark.gpu
SEED = synth.DEFAULT_SEED
ROOT = Path(__file__).resolve().parent.parent


def _cfg(**kw):
    base = dict(fft_size=2048, stereo=1, slope=1.0, bars=1, interp_mode=wf.INTERP["lanczos"])
    base.update(kw)
    return wf.Config.defaults(**base)


def _script(b, streams, ticks, hop, cc=2, gather=None):
    """the same calls on a SpectrumBatch or a MultiBatch: pushes that span shards, a hidden range that spans shards"""
    out = []
    for t in range(ticks):
        a = synth.block(SEED, 0, streams, cc, t * hop, hop)
        if t == 3:
            a[1:streams - 1] = 0.0
        b.push_audio(a[:2], first=0)
        b.push_audio(a[2:], first=2)
        if t == 4:
            m = np.zeros(streams, np.uint8)
            m[1:streams - 2] = 1
            b.set_hidden(m)
        if t == 6:
            b.set_hidden(np.zeros(streams - 1, np.uint8), first=1)
        b.tick()
        if gather is not None:
            gather(t)
        out.append((b.decibels(), b.bars(), b.last_silent()))
    return out


def test_one_device_group_is_bit_identical_to_a_plain_handle():
    cfg = _cfg()
    streams, ticks, hop = 9, 8, 800
    with wf.SpectrumBatch(cfg, streams) as plain:
        want = _script(plain, streams, ticks, hop)
    with wf.MultiBatch(cfg, streams, [0]) as m:
        assert m.n_devices == 1 and m.transport == "local" and m.shards[0][2:] == (0, streams)
        gathered = []

        def g(t):
            m.allgather_bars()
            gathered.append(m.gathered(0))
        got = _script(m, streams, ticks, hop, gather=g)
    for t, (w, g_) in enumerate(zip(want, got)):
        for a, c, what in zip(w, g_, ("rows", "bars", "m_last_silent")):
            assert np.array_equal(a, c), f"tick {t}: {what} of the one-device group differ from the plain handle"
        assert np.array_equal(gathered[t], w[1]), f"tick {t}: gathered bars differ"


@pytest.mark.parametrize("devices,streams", [([0, 0], 8), ([0, 0, 0], 7), ([0] * 5, 13)])
def test_shards_equal_one_handle_and_every_device_holds_every_bar(devices, streams):
    """several shards (on this box's one device, or cycling over all it has): rows, bars and silence flags equal the plain
    handle's; after the gather EVERY shard's device holds the bars of ALL streams in global order -- equal and ragged shards"""
    have = wf.device_count()
    devices = [(i % have) for i in range(len(devices))]
    cfg = _cfg()
    ticks, hop = 8, 800
    with wf.SpectrumBatch(cfg, streams) as plain:
        want = _script(plain, streams, ticks, hop)
    with wf.MultiBatch(cfg, streams, devices) as m:
        assert m.n_devices == len(devices)
        assert m.transport in ("peer", "rccl")
        assert sum(s[3] for s in m.shards) == streams and m.shards[0][2] == 0
        for a, c in zip(m.shards, m.shards[1:]):
            assert c[2] == a[2] + a[3] and 0 <= a[3] - c[3] <= 1
        gathered = []

        def g(t):
            m.allgather_bars()
            gathered.append([m.gathered(i) for i in range(m.n_devices)])
        got = _script(m, streams, ticks, hop, gather=g)
    for t, (w, g_) in enumerate(zip(want, got)):
        for a, c, what in zip(w, g_, ("rows", "bars", "m_last_silent")):
            assert np.array_equal(a, c), f"tick {t}: {what} of the sharded batch differ from the plain handle"
        for i, full in enumerate(gathered[t]):
            assert np.array_equal(full, w[1]), f"tick {t}: device index {i} holds different gathered bars"


def test_gather_runs_under_the_next_tick_and_is_double_buffered():
    """allgather_bars() after tick t does not wait; tick t+1 is issued right behind it; the result read afterwards is tick t's
    (not t+1's), on every device, over many back-to-back rounds (a send buffer is reused every second gather)"""
    cfg = _cfg(fft_size=4096)
    streams, ticks, hop = 384, 24, 800
    devices = [i % wf.device_count() for i in range(3)]
    with wf.SpectrumBatch(cfg, streams, ring_frames=4096 + hop * (ticks + 2)) as plain:
        plain.push_synth(SEED, 0, hop * (ticks + 1))
        per_tick = []
        for t in range(ticks + 1):
            plain.tick(delay_frames=hop * (ticks - t))
            per_tick.append(plain.bars())
    with wf.MultiBatch(cfg, streams, devices, ring_frames=4096 + hop * (ticks + 2)) as m:
        m.push_synth(SEED, 0, hop * (ticks + 1))
        m.tick(delay_frames=hop * ticks)
        for t in range(ticks):
            m.allgather_bars()                               # bars of tick t ...
            m.tick(delay_frames=hop * (ticks - 1 - t))       # ... while tick t+1 runs
            for i in range(m.n_devices):
                assert np.array_equal(m.gathered(i), per_tick[t]), f"gather {t} on device index {i}"
        assert np.array_equal(m.bars(), per_tick[ticks])
        # the timed loop: the same ticks + gathers driven by the devices' own threads
        ms, per = m.time_ticks(50, hop, hop * ticks, gather=True)
        assert ms > 0 and len(per) == m.n_devices and max(per) == pytest.approx(ms)
        last = [m.gathered(i) for i in range(m.n_devices)]
        assert all(np.array_equal(x, last[0]) for x in last) and np.array_equal(last[0], m.bars())


@pytest.mark.parametrize("n_fft,devices,mirror", [(4096, [0], "1"), (4096, [0], None), (4096, [0, 0, 0], None), (4096, [0, 0, 0], "send"), (800, [0, 0, 0], "1"),
                                                  (800, [0], None), (48000, [0, 0], None)])
def test_ticks_between_gathers_leave_the_gathered_result_alone(n_fft, devices, mirror, monkeypatch):
    """The contract of include/wf_hip.h: a gathered result stays valid until the FIRST tick after the NEXT gather -- whatever the
    number of ticks between two gathers, and whichever path the size takes: where the tick kernels write the send buffers / the
    results themselves (power-of-two sizes; round 5 switched buffers with every TICK, so the second tick after a gather overwrote
    the result being read) and where the bars are copied behind the tick (sizes that are not powers of two -- N = 800, the
    plugin's automatic size, mixed radix -- and sizes beyond a CU's LDS: wf_hip_set_bars_mirrors answers UNSUPPORTED and the group
    picks the copy for that handle).  `mirror`: WF_HIP_MULTI_MIRROR -- by default the kernels write only where that saves peer
    copies (every device addresses every other); "1" / "send" make them write the result / the send buffers too."""
    if mirror is None:
        monkeypatch.delenv("WF_HIP_MULTI_MIRROR", raising=False)
    else:
        monkeypatch.setenv("WF_HIP_MULTI_MIRROR", mirror)
    have = wf.device_count()
    devices = [(i % have) for i in range(len(devices))]
    cfg = _cfg(fft_size=n_fft)
    streams, hop = (24 if n_fft < 16384 else 6), 800
    pattern = [1, 2, 3, 1, 0, 2, 1]          # ticks in front of each gather (0: two gathers of the same tick)
    total = sum(pattern) + 4
    ring = n_fft + hop * (total + 2)
    with wf.SpectrumBatch(cfg, streams, ring_frames=ring) as plain, wf.MultiBatch(cfg, streams, devices, ring_frames=ring) as m:
        for b in (plain, m):
            b.push_synth(SEED, 0, hop * (total + 1))
        t = 0

        def tick():
            nonlocal t
            plain.tick(delay_frames=hop * (total - t)); m.tick(delay_frames=hop * (total - t))
            t += 1
        tick()  # (a first tick: nothing gathered yet)
        for k, n_ticks in enumerate(pattern):
            for _ in range(n_ticks):
                tick()
            want = plain.bars()
            m.allgather_bars()
            for i in range(m.n_devices):
                assert np.array_equal(m.gathered(i), want), f"gather {k} (after {n_ticks} ticks) on device index {i}"
            # ticks after the gather -- none of them may touch the result until another gather has been issued
            if k + 1 < len(pattern):
                for _ in range(pattern[k + 1]):
                    tick()
                    for i in range(m.n_devices):
                        assert np.array_equal(m.gathered(i), want), f"a tick behind gather {k} changed the gathered result on device index {i}"
                pattern[k + 1] = 0  # (already ticked)
        assert np.array_equal(m.bars(), plain.bars())


def test_a_gather_before_any_tick_and_shard_handles_ticked_directly():
    """a gather with no tick since the buffers were set hands over the handles' own bars (the reset state); shard handles ticked
    directly (the header allows it), one of them more often than the others, do not desynchronise the shards' buffers: the next
    gather is still every shard's newest bars, complete and the same on every device (round 5 derived the buffer from the
    shard's own tick count)"""
    import ctypes as C
    from waveform_amd import binding
    have = wf.device_count()
    devices = [(i % have) for i in range(3)]
    cfg = _cfg(fft_size=2048)
    streams, hop = 9, 800
    with wf.SpectrumBatch(cfg, streams) as plain, wf.MultiBatch(cfg, streams, devices) as m:
        m.allgather_bars()
        assert np.array_equal(m.gathered(0), plain.bars()) and np.array_equal(m.gathered(2), plain.bars())
        for t in range(5):
            a = synth.block(SEED, 0, streams, 2, t * hop, hop)
            m.push_audio(a)
            if t == 2:  # this frame every shard's handle is ticked by hand, shard 1 twice (a second tick of the same audio)
                p = binding.TickParams(1.0 / 60.0, 0, 0.0, 0, 0)
                for i, (h, dev, first, count) in enumerate(m.shards):
                    for _ in range(2 if i == 1 else 1):
                        assert m.L.wf_hip_tick(h, C.byref(p)) == 0
            else:
                m.tick()
            m.allgather_bars()
            got = [m.gathered(i) for i in range(m.n_devices)]
            assert all(np.array_equal(g, got[0]) for g in got), f"frame {t}: the devices hold different gathered results"
            assert np.array_equal(got[0], m.bars()), f"frame {t}: the gathered result is not the shards' newest bars"


def test_rccl_transport_in_a_fresh_process():
    """ncclAllGather through the dlopen()ed librccl.so: one rank per visible device (one on a 1-GPU box), equal and ragged
    shards; in a child process so that a broken RCCL cannot take the test session with it"""
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import waveform_amd as wf
from tools import synth
n = wf.device_count()
for streams in (8 * n, 8 * n + (1 if n > 1 else 0)):
    cfg = wf.Config.defaults(fft_size=2048, stereo=1, slope=1.0, bars=1, interp_mode=wf.INTERP["lanczos"])
    with wf.MultiBatch(cfg, streams, list(range(n))) as m:
        assert m.transport == "rccl", m.transport
        for t in range(6):
            m.push_audio(synth.block(synth.DEFAULT_SEED, 0, streams, 2, t * 800, 800))
            m.tick()
            m.allgather_bars()
        want = m.bars()
        for i in range(n):
            assert np.array_equal(m.gathered(i), want), (streams, i)
        ms, per = m.time_ticks(20, 0, 0, gather=True)
        assert np.array_equal(m.gathered(n - 1), m.bars())
    with wf.SpectrumBatch(cfg, streams) as p:
        for t in range(6):
            p.push_audio(synth.block(synth.DEFAULT_SEED, 0, streams, 2, t * 800, 800))
            p.tick()
        assert np.array_equal(p.bars(), want)
print("rccl ok", n)
''' % str(ROOT)
    env = dict(os.environ, WF_HIP_MULTI_TRANSPORT="rccl", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "rccl ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_every_visible_device_takes_a_shard():
    """n = wf_hip_device_count(): the default transport (RCCL when the devices are distinct and there is more than one) and the
    forced peer copies must agree with one plain handle -- runs with n = 1 on a 1-GPU box, n = 8 on a full node"""
    n = wf.device_count()
    cfg = _cfg(fft_size=1024)
    streams, ticks, hop = 4 * n + 3, 5, 800
    with wf.SpectrumBatch(cfg, streams) as plain:
        for t in range(ticks):
            plain.push_audio(synth.block(SEED, 0, streams, 2, t * hop, hop))
            plain.tick()
        want = (plain.decibels(), plain.bars())
    for force in (None, "peer"):
        old = os.environ.pop("WF_HIP_MULTI_TRANSPORT", None)
        if force:
            os.environ["WF_HIP_MULTI_TRANSPORT"] = force
        try:
            with wf.MultiBatch(cfg, streams, list(range(n))) as m:
                for t in range(ticks):
                    m.push_audio(synth.block(SEED, 0, streams, 2, t * hop, hop))
                    m.tick()
                m.allgather_bars()
                assert np.array_equal(m.decibels(), want[0]) and np.array_equal(m.bars(), want[1])
                for i in range(n):
                    assert np.array_equal(m.gathered(i), want[1]), (force, m.transport, i)
        finally:
            os.environ.pop("WF_HIP_MULTI_TRANSPORT", None)
            if old is not None:
                os.environ["WF_HIP_MULTI_TRANSPORT"] = old


@pytest.mark.parametrize("shards,transport,timed", [(3, "peer", False), (3, "peer", True), (1, "rccl", False), (1, "rccl", True)])
def test_a_failed_gather_leaves_a_working_group(shards, transport, timed):
    """One shard fails inside a gather while the others have their half of the exchange enqueued (test aid
    wf_hip_multi_debug_fail_next_gather): the call reports it, later gathers are refused, and nothing hangs -- the group goes on
    ticking, its results still equal a plain handle's, sync and destroy return.  Peer copies with three shards; ncclAllGather with
    the ranks this box has (ncclCommAbort is what releases the other ranks on a node).  In a child process with a timeout: the
    failure mode under test is a hang."""
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import waveform_amd as wf
from tools import synth
shards, timed = %d, %d
n = wf.device_count()
devices = [i %% n for i in range(shards)] if %r == "peer" else list(range(n))
cfg = wf.Config.defaults(fft_size=2048, stereo=1, slope=1.0, bars=1, interp_mode=wf.INTERP["lanczos"])
streams, hop = 12 * len(devices), 800
with wf.SpectrumBatch(cfg, streams, ring_frames=2048 + 40 * hop) as plain, wf.MultiBatch(cfg, streams, devices, ring_frames=2048 + 40 * hop) as m:
    assert m.transport == %r, m.transport
    for b in (plain, m):
        b.push_synth(synth.DEFAULT_SEED, 0, 30 * hop)
    for t in range(3):
        plain.tick(delay_frames=hop * (29 - t)); m.tick(delay_frames=hop * (29 - t)); m.allgather_bars()
    assert np.array_equal(m.gathered(0), plain.bars())
    assert m.L.wf_hip_multi_debug_fail_next_gather(m.m, len(devices) - 1) == 0
    try:
        if timed:
            m.time_ticks(6, hop, hop * 26, gather=True)
        else:
            m.tick(delay_frames=hop * 26); m.allgather_bars()
        raise SystemExit("the injected failure was not reported")
    except wf.WfHipError as e:
        assert "injected failure" in str(e), str(e)
    try:
        m.allgather_bars()
        raise SystemExit("a gather after the failure was accepted")
    except wf.WfHipError as e:
        assert "out of service" in str(e), str(e)
    m.sync()
    m.reset(); plain.reset()
    for b in (plain, m):
        b.push_synth(synth.DEFAULT_SEED, 0, 30 * hop)
    for t in range(5):
        plain.tick(delay_frames=hop * (29 - t)); m.tick(delay_frames=hop * (29 - t))
    assert np.array_equal(m.bars(), plain.bars()) and np.array_equal(m.decibels(), plain.decibels())
    ms, per = m.time_ticks(10, hop, hop * 20)
    assert ms > 0
print("survived")
''' % (str(ROOT), shards, int(timed), transport, transport)
    # (the test aid lives in the development build of the library only: libwaveform_hip_dev.so, the same kernel objects)
    env = dict(os.environ, WF_HIP_MULTI_TRANSPORT=transport, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               WF_HIP_LIB=str(ROOT / "waveform_amd" / "libwaveform_hip_dev.so"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "survived" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_multi_errors_are_reported():
    cfg = _cfg()
    with pytest.raises(wf.WfHipError) as e:
        wf.MultiBatch(cfg, 4, [wf.device_count()])     # no such device
    assert e.value.code == -1
    with pytest.raises(wf.WfHipError):
        wf.MultiBatch(cfg, 1, [0, 0])                   # fewer streams than shards
    with wf.MultiBatch(wf.Config.defaults(fft_size=1024), 4, [0, 0]) as m:   # no bars: nothing to gather
        with pytest.raises(wf.WfHipError):
            m.allgather_bars()
        with pytest.raises(wf.WfHipError):
            m.decibels(first=3, count=2)
        m.push_synth(SEED, 0, 800)
        m.tick()
        assert m.decibels().shape == (4, 2, 512)


def test_the_c_example_runs(tmp_path):
    """examples/multi_gpu_bars.c: the multi-device group from plain C, every device of the box, 120 ticks with the gather"""
    exe = tmp_path / "multi_gpu_bars"
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", f"-I{ROOT / 'include'}", str(ROOT / "examples" / "multi_gpu_bars.c"),
           f"-L{ROOT / 'waveform_amd'}", "-lwaveform_hip", "-lm", "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    env = dict(os.environ, LD_LIBRARY_PATH=f"{ROOT / 'waveform_amd'}:/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    run = subprocess.run([str(exe)], capture_output=True, text=True, env=env, timeout=600)
    assert run.returncode == 0 and "streams over" in run.stdout, (run.stdout, run.stderr[-2000:])


def test_node_check_runs_and_verifies():
    """tools/node_check.py -- what gets run first on a node nobody could rehearse on: every device; one process per leg (default
    transport, RCCL with one and two channels, RCCL with the gather stream on a queue of its own, the default transport with the
    tick kernels writing the send buffers, the peer transport with direct stores and with copies); per-device times with and without the gather, their difference, the gather alone, the links; exit
    code 0 only if every leg's gathered copies verified"""
    import json
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "node_check.py"), "--streams-per-device", "1024", "--ticks", "40"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["devices_used"] == wf.device_count() and len(out["runs"]) == 7
    # (legs: default, rccl x 3, the default transport with kernel-side stores, peer x 2)
    assert [run["transport"] for run in out["runs"]][1:4] == ["rccl", "rccl", "rccl"] and [run["transport"] for run in out["runs"]][5:] == ["peer", "peer"]
    for run in out["runs"]:
        assert run["verified"] and run["gather_alone_us"] > 0 and len(run["ms_per_tick_with_gather"]["per_device"]) == len(run["devices"]), run
        assert "gather_costs_per_tick_us" in run and "leg" in run


def test_node_check_runs_on_this_box():
    """tools/node_check.py, the first thing to run on a multi-GPU node: here with the devices this box has (the peer legs put two
    shards on one device where there is one) and the host-fed leg -- exit code 0, "ready", no reason lines, a GB/s figure per device.
    (Its verdicts -- RCCL refuses the device list, peer access denied, a slow device -- are unit-tested on fabricated results in
    tests/test_cpu_units.py::test_node_check_names_every_reason.)"""
    import json
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "node_check.py"), "--ticks", "20", "--streams-per-device", "512", "--host-fed", "--host-fed-streams", "256"],
                       capture_output=True, text=True, timeout=900)
    line = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert line, (r.stdout[-500:], r.stderr[-2000:])
    d = json.loads(line[-1])
    assert r.returncode == 0 and d["ready"] and d["reasons"] == [], (r.returncode, d.get("reasons"), r.stderr[-1500:])
    assert len(d["runs"]) == 7 and all(x.get("verified") for x in d["runs"])
    hf = d["host_fed"]
    assert hf["verified"] and len(hf["per_device"]) == d["devices_used"] and all(x["host_GBps"] > 0 for x in hf["per_device"])
