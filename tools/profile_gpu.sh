#!/bin/bash
# tools/profile_gpu.sh -- rocprofv3 passes over bench.py on the GPU box (run through gpurun).
#   pass 1: --kernel-trace --stats          -> per-kernel durations
#   pass 2..: --pmc <counters> (no traces)  -> HBM bytes (FETCH_SIZE / WRITE_SIZE in separate passes), SQ/LDS counters
# Outputs land in gpurun_out/prof/<tag>/ ; summaries worth keeping are copied to profiles/ by hand.
set -u
TAG=${1:-r01}
STEPS=${2:-30}
EXTRA=${3:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
export WF_HIP_ROCTX=1   # roctx ranges around every wf_hip_tick (resolved from the profiler's marker library at run time)
CMD=${WF_PROFILE_CMD:-"python $R/bench.py --steps $STEPS --warmup 3 --no-cpu-baseline --no-other-configs $EXTRA"}   # WF_PROFILE_CMD: profile another driver (e.g. tools/meter_bench.py)
echo "$CMD" | sed "s#$R/##g" > $OUT/cmd.txt
rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d $OUT/stats -o stats -- $CMD > $OUT/stats.log 2>&1   # (--marker-trace: the roctx range around every wf_hip_tick)
if [ "${WF_PMC_SET:-full}" = "short" ]; then   # HBM bytes only
for PMC in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc_$PMC -o pmc -- $CMD > $OUT/pmc_$PMC.log 2>&1
done
find $OUT -name "*.csv" | head -40
exit 0
fi
for PMC in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT"; do
  NAME=$(echo $PMC | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc_$NAME -o pmc -- $CMD > $OUT/pmc_$NAME.log 2>&1
done
find $OUT -name "*.csv" | head -40
