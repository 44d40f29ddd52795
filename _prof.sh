cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats -o s -- python3 bench.py --steps 30 --warmup 8 > gpurun_out/stats_bench.log 2>&1
tail -1 gpurun_out/stats_bench.log | cut -c1-200
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/stats/s_kernel_stats.csv')))
for r in rows[:26]:
    print(f"{r['Name'][:70]:70s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.1f} us {r['Percentage']:>6s}%")
PY
