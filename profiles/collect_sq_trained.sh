#!/bin/bash
# SQ counters of the step's kernels on the TRAINED scene (bench.py --trained-only): bash profiles/collect_sq_trained.sh r03
# The per-kernel mean is taken over the second half of the dispatches, i.e. over steps 750-1500 of the fit and the timed
# steps after it.  Output: gpurun_out/prof_<round>/sq_counters_trained.csv
set -e
R=${1:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/sqt_$i -o p -- python3 bench.py --steps 6 --warmup 3 --trained-only --trained-steps 1500 --no-cpu-baseline > $OUT/sqt_$i.log 2>&1
done
python3 profiles/summarize_sq.py $OUT/sqt_1/p_counter_collection.csv $OUT/sqt_2/p_counter_collection.csv > $OUT/sq_counters_trained.csv
rm -rf $OUT/sqt_1/p_kernel_trace.csv $OUT/sqt_2/p_kernel_trace.csv $OUT/sqt_1/p_counter_collection.csv $OUT/sqt_2/p_counter_collection.csv
cat $OUT/sq_counters_trained.csv
