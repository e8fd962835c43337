#!/bin/bash
# tools/gpu_round.sh -- one gpurun call that gathers a round's evidence: GPU tests, bench line, rocprofv3 passes.
#   usage (from the repo root on the GPU box): bash tools/gpu_round.sh TAG [tests|notests]
set -u
TAG=${1:-r02k}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
if [ "${2:-tests}" = "tests" ]; then
  ( time timeout 1800 python -m pytest tests -m gpu -x -q --durations=15 -n ${WF_XDIST:-0} ) > $O/gputests_$TAG.log 2>&1
  tail -4 $O/gputests_$TAG.log
fi
( time python bench.py ) > $O/bench_$TAG.json 2> $O/bench_$TAG.err
tail -c 600 $O/bench_$TAG.json | head -c 600; echo
# headline shape: full counter set; the other shapes: durations + HBM bytes
python tools/warmup_curve.py 40 > $O/warmup_$TAG.txt 2>&1
bash tools/profile_gpu.sh ${TAG}_cfg3 1000 "--warmup 500" > /dev/null 2>&1   # the default bench.py run: same ring depth, same ticks
WF_PMC_SET=short WF_PROFILE_CMD="python $R/tools/shape_bench.py cfg3_8192streams" bash tools/profile_gpu.sh ${TAG}_cfg3_8192streams > /dev/null 2>&1
WF_PMC_SET=short bash tools/profile_gpu.sh ${TAG}_cfg3_16384streams 250 "--streams 16384 --warmup 250" > /dev/null 2>&1
WF_PMC_SET=short WF_PROFILE_CMD="python $R/tools/shape_bench.py cfg2_batch" bash tools/profile_gpu.sh ${TAG}_cfg2 > /dev/null 2>&1
WF_PMC_SET=short WF_PROFILE_CMD="python $R/tools/shape_bench.py cfg4_n16384_bars" bash tools/profile_gpu.sh ${TAG}_cfg4 > /dev/null 2>&1
WF_PMC_SET=short WF_PROFILE_CMD="python $R/tools/shape_bench.py cfg5shape_8192streams_barsonly" bash tools/profile_gpu.sh ${TAG}_cfg5shape > /dev/null 2>&1
# the kernels further from the roofline: durations (rocprofv3 --kernel-trace --stats) and HBM bytes (FETCH_SIZE / WRITE_SIZE in
# passes of their own), one summary each
for J in "plugindefaults:python $R/tools/quick_case.py plugin_defaults" "n32768:python $R/tools/quick_bench.py 32768:512" \
         "n800:python $R/tools/quick_bench.py 800:8192" "n4160:python $R/tools/quick_bench.py 4160:2048" \
         "n65536:python $R/tools/quick_bench.py 65536:256" \
         "n16400:python $R/tools/quick_bench.py 16400:512" "n48000:python $R/tools/quick_bench.py 48000:256" "n48016:python $R/tools/quick_bench.py 48016:256" "n32000:python $R/tools/quick_bench.py 32000:256" \
         "meter:python $R/tools/meter_bench.py" "wave:python $R/tools/wave_bench.py"; do
  NAME=${J%%:*}; CMD=${J#*:}
  WF_PMC_SET=short WF_PROFILE_CMD="$CMD" bash tools/profile_gpu.sh ${TAG}_$NAME > /dev/null 2>&1
done
bash tools/size_sweep.sh $TAG > /dev/null 2>&1
python tools/node_check.py > $O/node_check_$TAG.json 2> $O/node_check_$TAG.err
ls $O/prof
# The raw rocprofv3 output of seventeen shapes exceeds what gpurun copies back (64 MiB): summarised HERE into profiles/TAG_* (the files
# that get committed), which travel home in gpurun_out/profiles_TAG/; the raw passes stay on the box.
python - <<PY
import json
lines = [l for l in open("$O/bench_$TAG.json").read().splitlines() if l.startswith("{")]
if lines:
    open("profiles/${TAG}_bench_line.json", "w").write(lines[-1] + "\n")
PY
bash tools/summarize_round.sh $TAG
python tools/profiles_summary.py $TAG > /dev/null 2>&1
mkdir -p $O/profiles_$TAG
cp profiles/${TAG}_* $O/profiles_$TAG/ 2>/dev/null
rm -rf $O/prof
du -sh $O
