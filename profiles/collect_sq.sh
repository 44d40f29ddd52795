#!/bin/bash
# SQ counters of the step's kernels (evidence for "VALU-issue-bound"): bash profiles/collect_sq.sh r01  (through gpurun)
# Counters only with --kernel-trace, a few per pass.  Output: gpurun_out/prof_<round>/sq_counters.csv
set -e
R=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/sq_$i -o p -- python3 bench.py --steps 6 --warmup 3 --no-extras --no-cpu-baseline > $OUT/sq_$i.log 2>&1
done
python3 profiles/summarize_sq.py $OUT/sq_1/p_counter_collection.csv $OUT/sq_2/p_counter_collection.csv > $OUT/sq_counters.csv
rm -rf $OUT/sq_1/p_kernel_trace.csv $OUT/sq_2/p_kernel_trace.csv
cat $OUT/sq_counters.csv
