"""Every position of the reference's FFT-size slider above 16384 (src/source.cpp:359-363: 64 ... 65536 in steps of 64 -- 768
positions from 16448 to 65536), each one (1) played through a drawn fuzz scenario (tests/test_gpu_fuzz.py's "huge" family with the
size forced) against the reference itself, oracle/_ref/libwfref.so, and (2) timed: 256 stereo streams, back-to-back ticks, steady
state (tools/quick_bench.py's method with shorter regions).  One JSON line per position:
    python tests/sizes_large_sweep.py [OUT.jsonl [LO [HI [STEP [parity | STREAMS]]]]]   (development aid / evidence: profiles/r05_sizes_large.jsonl)
With "parity" as the fifth argument the timing is left out and the lines are short -- the form used for EVERY legal size, the 4089
multiples of 16 from 128 to 65536 (profiles/r05_sizes_all_parity.jsonl: python tests/sizes_large_sweep.py OUT 128 65536 16 parity).
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz as f  # noqa: E402
import waveform_amd as wf  # noqa: E402
from tools import synth, quick_bench  # noqa: E402

out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r05_sizes_large.jsonl")
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 16384 + 64
hi = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
step = int(sys.argv[4]) if len(sys.argv) > 4 else 64
parity_only = len(sys.argv) > 5 and sys.argv[5] == "parity"
STREAMS, HOP, TICKS = 256, 800, 12
if len(sys.argv) > 5 and sys.argv[5].isdigit():  # (another batch size: the sizes up to 16384 want more streams to fill the chip)
    STREAMS = int(sys.argv[5])
quick_bench.WARM_MS, quick_bench.TIMED_MS = 12.0, 8.0


def family(name: str) -> str:
    for key, fam in (("big_mr_whole", "mixed radix: two rows in one kernel"), ("big_mr_rows", "mixed-radix rows"), ("big_br_", "Bluestein rows in LDS"), ("big_whole", "one kernel (65536)"),
                     ("big_{columns", "through device memory"), ("spectrum_tick_kernel", "fused tick kernel")):
        if key in name:
            return fam
    return "?"


def timed(n: int) -> dict:
    cfg = wf.Config.defaults(fft_size=n, stereo=1, slope=1.0)
    with wf.SpectrumBatch(cfg, STREAMS, ring_frames=n + HOP * (TICKS + 4)) as b:
        b.push_synth(synth.DEFAULT_SEED, 0, HOP * (TICKS + 2))
        b.sync()
        ms = quick_bench.steady_ms(lambda k: b.time_ticks(k, HOP, HOP * (TICKS - 1)), TICKS, 2)
        byt = b.algorithmic_bytes_per_tick()
        return dict(kernel=b.kernel_name(), ms=round(ms, 4), algorithmic_bytes_per_tick=int(byt), frac=round(byt / ms / 1e6 / 8000, 4))


if not parity_only:
    quick_bench.warm_clocks()
bad = 0
with open(out_path, "w") as out:
    for i, n in enumerate(range(lo, hi + 1, step)):
        rec = dict(fft_size=n) if parity_only else dict(fft_size=n, streams=STREAMS)
        try:
            f.run_spectrum_case(i, "huge" if n > 16384 else "any", fft_size=n)
            rec["parity"] = "ok"
        except Exception as e:  # noqa: BLE001 -- the sweep records and goes on
            bad += 1
            rec["parity"] = "FAIL: " + str(e).replace("\n", " ")[-300:]
        try:
            if parity_only:
                with wf.SpectrumBatch(wf.Config.defaults(fft_size=n), 1) as b:
                    rec["path"] = family(b.kernel_name()) + (": mixed radix" if "mixed radix" in b.kernel_name() and n <= 16384 else ": Bluestein" if "Bluestein" in b.kernel_name() and n <= 16384 else "")
            else:
                rec.update(timed(n))
                rec["path"] = family(rec["kernel"])
        except Exception as e:  # noqa: BLE001
            rec["timing_error"] = str(e)[-200:]
        out.write(json.dumps(rec) + "\n")
        out.flush()
print("positions", (hi - lo) // step + 1, "parity failures", bad, "unsupported", len(f.ARM["unsupported"]))
