#!/bin/bash
# A/B runs of kernel variants on the GPU box: every library under profiles/_bin/variants/<name>/libw3d_hip.so (built by
# profiles/build_variant.sh; git-ignored scratch) and the product library are benchmarked back to back with bench.py
# (headline + trained scene, per-stage event times); one line per variant lands in gpurun_out/ab.log.
#   usage: profiles/ab_variants.sh [steps] [trained_steps]
cd "$(dirname "$0")/.."
STEPS=${1:-60}; TR=${2:-1500}
mkdir -p gpurun_out
: > gpurun_out/ab.log
for lib in wheat-3dgs_amd/lib/libw3d_hip.so profiles/_bin/variants/*/libw3d_hip.so; do
  [ -f "$lib" ] || continue
  name=$(basename "$(dirname "$lib")")
  W3D_HIP_LIB=$lib timeout 300 python bench.py --steps $STEPS --warmup 10 --trained-only --trained-steps $TR --no-cpu-baseline 2>/dev/null \
    | python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); t=d.get('trained_scene') or {}
        print('$name', 'it/s', d['value'], 'trained', t.get('value'), 'stages', json.dumps(d['stage_ms']), 'trained_stages', json.dumps(t.get('stage_ms')))
" >> gpurun_out/ab.log
done
cat gpurun_out/ab.log
