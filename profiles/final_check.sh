#!/bin/bash
timeout 2700 python -m pytest tests -q -m gpu > gpurun_out/r06_gpu_suite.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" gpurun_out/r06_gpu_suite.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time python bench.py ) > gpurun_out/r06_bench_final.json 2> gpurun_out/r06_bench_final.err; grep real gpurun_out/r06_bench_final.err; wc -c gpurun_out/r06_bench_final.json; python3 -c "
import json; d=json.loads(open('gpurun_out/r06_bench_final.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['parity']['unattributed_outliers'], d['parity']['radii_differing'], d['cpu_baseline']['value'])"
