#!/bin/bash
# Collects the round's profile artefacts on the GPU box (run through gpurun from the repo root):
#   bash profiles/collect.sh r02
# 1. bench line (100 steps)                       -> gpurun_out/prof_<round>/bench_n1_s100.json
# 2. rocprofv3 --kernel-trace --stats (30 steps)  -> gpurun_out/prof_<round>/bench_s30_kernel_stats.csv
# 2b. the same for the unmodified reference loop alone (bench.py --dropin-only)  -> dropin_s40_kernel_stats.csv
# 3. PMC passes, counters only with --kernel-trace (FETCH_SIZE, WRITE_SIZE in separate runs)
#                                                 -> gpurun_out/prof_<round>/pmc_hbm_traffic.csv
# Copy the three files into profiles/<round>/ afterwards (gpurun_out/ is scratch).
set -e
R=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py --steps 100 --warmup 20 2>$OUT/bench.err | tail -1 > $OUT/bench_n1_s100.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 bench.py --steps 30 --warmup 8 --no-extras --no-cpu-baseline > $OUT/stats.log 2>&1
cp $OUT/stats/s_kernel_stats.csv $OUT/bench_s30_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dstats -o s -- python3 bench.py --steps 40 --warmup 8 --dropin-only > $OUT/dstats.log 2>&1
cp $OUT/dstats/s_kernel_stats.csv $OUT/dropin_s40_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o p -- python3 bench.py --steps 6 --warmup 3 --no-extras --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o p -- python3 bench.py --steps 6 --warmup 3 --no-extras --no-cpu-baseline > $OUT/pmc_write.log 2>&1
python3 profiles/summarize_pmc.py $OUT/pmc_fetch/p_counter_collection.csv $OUT/pmc_write/p_counter_collection.csv > $OUT/pmc_hbm_traffic.csv
rm -rf $OUT/dstats/s_kernel_trace.csv $OUT/stats/s_kernel_trace.csv $OUT/pmc_fetch/p_kernel_trace.csv $OUT/pmc_write/p_kernel_trace.csv
cat $OUT/bench_n1_s100.json | cut -c1-400
head -12 $OUT/pmc_hbm_traffic.csv
