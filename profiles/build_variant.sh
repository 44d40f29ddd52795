#!/bin/bash
# profiles/build_variant.sh <name> <file.hip> [-DFLAG=...]...: the product library with ONE translation unit rebuilt with
# extra flags, into profiles/_bin/variants/<name>/ (scratch: git-ignored, never the product path).
set -e
cd "$(dirname "$0")/.."
name=$1; unit=$2; shift 2
out=profiles/_bin/variants/$name
mkdir -p $out
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -munsafe-fp-atomics -fno-slp-vectorize -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS "$@" -c wheat-3dgs_amd/csrc/$unit.hip -o $out/$unit.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libw3d_hip.so $out/$unit.o $(ls wheat-3dgs_amd/lib/w3d_*.o | grep -v "/$unit.o")
echo "built $out/libw3d_hip.so"
