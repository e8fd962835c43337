#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the reference itself (oracle/_ref/libwfref.so).

Run in the build container (where /root/reference exists and `make -C oracle/ref` has been
run).  Every fixture holds the outputs the reference produced for one scenario of
tests/scenarios.py; inputs are regenerated from the counter hash, so fixtures stay small and
can travel to the GPU box, where /root/reference does not exist.

    python tools/make_golden.py            # (re)write all fixtures
    python tools/make_golden.py NAME ...   # only the named scenarios
"""
from __future__ import annotations

import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import scenarios  # noqa: E402


def main():
    out_dir = ROOT / "tests" / "golden"
    out_dir.mkdir(parents=True, exist_ok=True)
    total = 0
    only = set(sys.argv[1:])
    unknown = only - set(scenarios.SCENARIOS)
    if unknown:
        raise SystemExit(f"unknown scenarios: {sorted(unknown)}")
    for name, sc in scenarios.SCENARIOS.items():
        if only and name not in only:
            continue
        cfg = scenarios.make_config(sc["cfg"])
        be = scenarios.RefBackend(cfg, isa="generic")
        recs = scenarios.play(be, sc)
        arrays = {}
        silent = np.array([r["silent"] for r in recs], np.uint8)
        if recs and "rms" in recs[0]:
            arrays["rms"] = np.array([r["rms"] for r in recs], np.float32)  # m_input_rms after every tick
        for t, r in scenarios.recorded(recs, sc["record"]):
            arrays[f"db_{t}"] = r["db"].astype(np.float32)
            if r["bars"] is not None:
                arrays[f"bars_{t}"] = r["bars"].astype(np.float32)
            if "verts" in r:
                for c, v in enumerate(r["verts"]):
                    arrays[f"verts_{t}_c{c}"] = v.astype(np.float32)
        meta = dict(scenario=name, cfg=sc["cfg"], n_ticks=len(recs), generator="tools/make_golden.py",
                    source="oracle/_ref/libwfref.so = phandasm/waveform v1.9.1 WAVSourceGeneric + vendored FFTW 3.3.11")
        p = out_dir / f"{name}.npz"
        np.savez_compressed(p, silent=silent, meta=np.frombuffer(json.dumps(meta).encode(), np.uint8), **arrays)
        total += p.stat().st_size
        print(f"{name:32s} ticks={len(recs):3d} recorded={len(arrays):3d} arrays  {p.stat().st_size/1024:7.1f} KiB")
    print(f"total {total/1024:.1f} KiB")


if __name__ == "__main__":
    main()
