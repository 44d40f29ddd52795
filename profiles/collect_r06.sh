#!/bin/bash
# Round 6 profile collection (one gpurun call): bash profiles/collect_r06.sh   -> gpurun_out/prof_r06/, copy the summaries into profiles/r06/
#   bench_n1.json + bench_detail.json        the default `python bench.py` line and its detail file
#   bench_full.json + bench_detail_full.json `python bench.py --full` (every leg)
#   bench_kernel_stats.csv                   rocprofv3 --kernel-trace --stats of `bench.py --steps 30 --no-cpu-baseline` (the same command, fewer steps)
#   {kernel_stats,pmc_hbm_traffic,sq_counters}_<scene>.csv   per scene (profiles/collect_scenes.sh: marker window over profiles/scene_step.py)
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py --detail-file $OUT/bench_detail.json 2>$OUT/bench.err | tail -1 > $OUT/bench_n1.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --detail-file none > $OUT/stats.log 2>&1
cp $OUT/stats/s_kernel_stats.csv $OUT/bench_kernel_stats.csv; rm -rf $OUT/stats
bash profiles/collect_scenes.sh r06 "untrained trained densified opaque" > $OUT/collect_scenes.log 2>&1 || tail -5 $OUT/collect_scenes.log
python3 bench.py --full --detail-file $OUT/bench_detail_full.json 2>$OUT/bench_full.err | tail -1 > $OUT/bench_full.json
ls -la $OUT
