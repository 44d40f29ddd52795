#!/bin/bash
# Round 5's profile artefacts in one gpurun call: bash profiles/collect_r05.sh   -> gpurun_out/prof_r05/ (copy into profiles/r05/)
#   bench_n1.json                      the default `python bench.py` line
#   bench_s30_kernel_stats.csv         rocprofv3 --kernel-trace --stats of bench.py --steps 30 (headline step only)
#   dropin_s40_kernel_stats.csv        the unmodified loop script under the import redirect (bench.py --dropin-only)
#   {kernel_stats,pmc_hbm_traffic,sq_counters}_<scene>.csv for the four scenes (profiles/collect_scenes.sh)
set -e
R=r05
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py 2>$OUT/bench.err | tail -1 > $OUT/bench_n1.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 bench.py --steps 30 --warmup 8 --no-extras --no-cpu-baseline > $OUT/stats.log 2>&1
cp $OUT/stats/s_kernel_stats.csv $OUT/bench_s30_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dstats -o s -- python3 bench.py --steps 40 --warmup 8 --dropin-only > $OUT/dstats.log 2>&1
cp $OUT/dstats/s_kernel_stats.csv $OUT/dropin_s40_kernel_stats.csv
rm -rf $OUT/stats $OUT/dstats
bash profiles/collect_scenes.sh $R "untrained trained densified opaque"
cut -c1-600 $OUT/bench_n1.json
