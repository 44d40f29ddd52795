#!/bin/bash
# profiles/build_variant.sh <name> <unit|all> [-DFLAG=...]...: the product library with ONE translation unit (or, with `all`,
# every unit — for macros of w3d_common.h) rebuilt with extra flags, into profiles/_bin/variants/<name>/ (scratch: git-ignored,
# never the product path).
set -e
cd "$(dirname "$0")/.."
name=$1; unit=$2; shift 2
out=profiles/_bin/variants/$name
mkdir -p $out
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -munsafe-fp-atomics -fno-slp-vectorize -Wall -Wno-unused-function"
if [ "$unit" = all ]; then
  pids=()
  for f in wheat-3dgs_amd/csrc/w3d_*.hip; do
    /opt/rocm/bin/hipcc $FLAGS "$@" -c $f -o $out/$(basename $f .hip).o &
    pids+=($!)
  done
  for p in "${pids[@]}"; do wait $p; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libw3d_hip.so $out/w3d_*.o
else
  /opt/rocm/bin/hipcc $FLAGS "$@" -c wheat-3dgs_amd/csrc/$unit.hip -o $out/$unit.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libw3d_hip.so $out/$unit.o $(ls wheat-3dgs_amd/lib/w3d_*.o | grep -v "/$unit.o")
fi
echo "built $out/libw3d_hip.so"
