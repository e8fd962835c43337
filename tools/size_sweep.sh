#!/bin/bash
# tools/size_sweep.sh TAG -- device time per tick of every FFT size family (tools/quick_bench.py: 30 back-to-back ticks over
# shallow rings, HIP events; best of 3) into gpurun_out/<TAG>_sizes.jsonl.  Numbers in DESIGN.md's size table come from here;
# they are event timings, not rocprofv3 profiles.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/${1:-r02}_sizes.jsonl; mkdir -p gpurun_out; : > $O
for S in 128:16384 256:16384 512:16384 1024:16384 2048:8192 4096:4096 8192:2048 16384:1024 32768:512 65536:256 800:8192 800:32768 960:8192 1600:8192 4160:2048 8000:1024 16400:256 32000:256 48000:256 48016:256 65424:256; do
  python tools/quick_bench.py $S 2>/dev/null >> $O; done
for S in 1024:16384 2048:8192 4096:4096 8192:2048 16384:1024 32768:512 65536:256; do
  WF_BENCH_BARS=1 python tools/quick_bench.py $S 2>/dev/null | sed 's/^{/{"bars": "26 Lanczos bars per row", /' >> $O; done
python - >> $O 2>/dev/null <<'PY'
import sys
sys.path.insert(0, '.')
from tools import quick_bench as q
print('{"note": "next three: plugin defaults (mono mixdown of 2 channels, 800-point Catmull-Rom curve); the same without the curve; stereo + 800-point Lanczos curve"}')
q.run(4096, 4096, stereo=0, curve=1, interp_mode=2)
q.run(4096, 4096, stereo=0)
q.run(4096, 4096, stereo=1, curve=1, interp_mode=1)
PY
python tools/meter_bench.py > gpurun_out/${1:-r02}_meter.txt 2>&1
python tools/wave_bench.py > gpurun_out/${1:-r02}_wave.txt 2>&1
