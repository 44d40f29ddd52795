#!/bin/bash
# profiles/build_ref.sh <name> <git-ref>: the product library as it was at <git-ref>, into profiles/_bin/variants/<name>/ — the "before"
# side of a same-box A/B (profiles/ab_scenes.sh, profiles/kstats_libs.sh run every library under profiles/_bin/variants/).
set -e
cd "$(dirname "$0")/.."
name=$1; ref=$2
out=$PWD/profiles/_bin/variants/$name
src=/tmp/w3d_ref_$name
rm -rf $src; mkdir -p $src/wheat-3dgs_amd/csrc $src/include $out
git archive $ref wheat-3dgs_amd/csrc include | tar -x -C $src
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -munsafe-fp-atomics -fno-slp-vectorize -Wall -Wno-unused-function"
pids=()
for f in $src/wheat-3dgs_amd/csrc/w3d_*.hip; do
  /opt/rocm/bin/hipcc $FLAGS -c $f -o $out/$(basename $f .hip).o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libw3d_hip.so $out/w3d_*.o
echo "built $out/libw3d_hip.so from $ref"
