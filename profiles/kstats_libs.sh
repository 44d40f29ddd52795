#!/bin/bash
# Per-kernel average durations (rocprofv3 --kernel-trace --stats) of K steps of one bench scene, for the product library and every
# profiles/_bin/variants/<name>/libw3d_hip.so, in ONE gpurun call (boxes differ: variants are only compared within a call).
#   usage: profiles/kstats_libs.sh <kernel-name substring> [scene] [steps]      -> gpurun_out/kstats_libs.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PAT=${1:-depth_}; S=${2:-untrained}; STEPS=${3:-60}
mkdir -p gpurun_out
rm -f /tmp/w3d_kl_$S.pt
python3 profiles/scene_step.py --scene $S --steps 2 --model-file /tmp/w3d_kl_$S.pt > /dev/null 2>&1
: > gpurun_out/kstats_libs.txt
for lib in wheat-3dgs_amd/lib/libw3d_hip.so profiles/_bin/variants/*/libw3d_hip.so; do
  [ -f "$lib" ] || continue
  name=$(basename $(dirname $lib)); rm -rf /tmp/kl_$name
  export W3D_HIP_LIB=$GRAFT_REPO_ROOT/$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kl_$name -o s -- python3 profiles/scene_step.py --scene $S --steps $STEPS --model-file /tmp/w3d_kl_$S.pt > /tmp/kl_$name.log 2>&1
  python3 - "$name" "$PAT" /tmp/kl_$name/s_kernel_stats.csv <<'PY' | tee -a gpurun_out/kstats_libs.txt
import csv, sys
name, pat, f = sys.argv[1:4]
tot = 0.0
for r in csv.DictReader(open(f)):
    if pat in r["Name"]:
        n = r["Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:48]
        print(f"{name:16s} {n:48s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}")
        tot += float(r["AverageNs"]) / 1e3
print(f"{name:16s} sum of averages {tot:8.1f} us")
PY
done
