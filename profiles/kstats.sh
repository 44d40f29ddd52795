#!/bin/bash
# rocprofv3 --kernel-trace --stats of the benchmark step (headline scene; with TRAINED=1 also the scene after 1500 training steps,
# whose kernels then dominate the averages) -> gpurun_out/kstats_<tag>.csv
#   usage: profiles/kstats.sh <tag> [extra bench.py flags]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/kstats_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kstats_$tag -o s -- python3 bench.py --steps 30 --warmup 8 --no-extras --no-cpu-baseline "$@" > gpurun_out/kstats_$tag.log 2>&1
cp gpurun_out/kstats_$tag/s_kernel_stats.csv gpurun_out/kstats_$tag.csv
rm -rf gpurun_out/kstats_$tag
python3 - <<PY
import csv
for r in list(csv.DictReader(open("gpurun_out/kstats_$tag.csv")))[:22]:
    n = r["Name"].replace("void (anonymous namespace)::", "").split("(")[0][:60]
    print(f"{n:60s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:9.1f} us {r['Percentage']:>6s}%")
PY
