#!/bin/bash
# tools/valu_per_wave.sh SHAPE lib.so ... -- VALU / SALU / LDS / VMEM instructions per wavefront of the tick kernel, one rocprofv3
# counter pass per library build; WF_VPW_CMD="python tools/shape_bench.py cfg5shape_8192streams_barsonly" replaces the quick_bench command (development aid: what a phase costs, with the -DWF_EXP_CUT_AT builds of tools/variant.sh)
SHAPE=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
for LIB in "$@"; do
  D=/tmp/vpw_$(basename $LIB .so)
  rm -rf $D
  WF_HIP_LIB=$R/$LIB rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d $D -o pmc -- ${WF_VPW_CMD:-python $R/tools/quick_bench.py $SHAPE} > $D.log 2>&1
  python3 - $D $LIB <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/pmc_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "tick_kernel" in r["Kernel_Name"] or "big_" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
w = m.get("SQ_WAVES", 0) or 1
print(sys.argv[2], {k: round(v / w, 1) for k, v in m.items() if k != "SQ_WAVES"}, "waves", w)
PY
done
