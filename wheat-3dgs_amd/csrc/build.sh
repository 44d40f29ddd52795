#!/bin/bash
# Builds libw3d_hip.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")"
mkdir -p ../lib
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -munsafe-fp-atomics -fno-slp-vectorize -Wall -Wno-unused-function"
pids=()
for f in w3d_preprocess w3d_binning w3d_render w3d_knn w3d_api w3d_loss w3d_adam w3d_densify w3d_mask w3d_exchange; do
  [ -f $f.hip ] || continue
  if [ ! -f ../lib/$f.o ] || [ $f.hip -nt ../lib/$f.o ] || [ w3d_common.h -nt ../lib/$f.o ] || [ ../../include/w3d.h -nt ../lib/$f.o ]; then
    $HIPCC $FLAGS "$@" -c $f.hip -o ../lib/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../lib/libw3d_hip.so ../lib/w3d_*.o
echo "built $(cd ../lib && pwd)/libw3d_hip.so"
