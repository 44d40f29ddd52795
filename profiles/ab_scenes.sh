#!/bin/bash
# A/B of kernel variants on bench.py's three scenes in ONE gpurun call (boxes differ by up to 15 % on the latency-bound kernels:
# variants are only ever compared within one call).  The scenes are prepared once with the product library and saved; then the
# product library and every profiles/_bin/variants/<name>/libw3d_hip.so (profiles/build_variant.sh) time the same models.
#   usage: profiles/ab_scenes.sh [steps] ["scenes"] [rounds]      -> gpurun_out/ab_scenes.log (one JSON line per lib, scene, round)
cd "$(dirname "$0")/.."
STEPS=${1:-60}; SCENES=${2:-"untrained trained densified"}; ROUNDS=${3:-2}; EXTRA=${4:-}     # EXTRA: e.g. --freeze
mkdir -p gpurun_out
: > gpurun_out/ab_scenes.log
for S in $SCENES; do
  rm -f /tmp/w3d_ab_$S.pt
  python3 profiles/scene_step.py --scene $S --steps 2 --model-file /tmp/w3d_ab_$S.pt > /dev/null 2>&1
done
for r in $(seq 1 $ROUNDS); do
  for lib in wheat-3dgs_amd/lib/libw3d_hip.so profiles/_bin/variants/*/libw3d_hip.so; do
    [ -f "$lib" ] || continue
    for S in $SCENES; do
      W3D_HIP_LIB=$lib timeout 300 python3 profiles/scene_step.py --scene $S --steps $STEPS --model-file /tmp/w3d_ab_$S.pt --report $EXTRA 2>/dev/null \
        | grep '^{' | sed "s#\"lib\": \"[^\"]*\"#\"lib\": \"$(basename $(dirname $lib))\"#" >> gpurun_out/ab_scenes.log
    done
  done
done
python3 - <<'PY'
import json, collections
rows = [json.loads(l) for l in open("gpurun_out/ab_scenes.log")]
acc = collections.defaultdict(list)
for r in rows:
    acc[(r["scene"], r["lib"])].append(r)
for (scene, lib), rs in sorted(acc.items()):
    ms = sorted(x["ms_per_step"] for x in rs)
    st = {k: round(sum(x["stage_ms"][k] for x in rs) / len(rs), 4) for k in rs[0]["stage_ms"]}
    print(f"{scene:10s} {lib:28s} ms/step {ms}  " + " ".join(f"{k}={v}" for k, v in sorted(st.items())))
PY
