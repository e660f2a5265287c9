#!/bin/bash
# Build profiling variants of the library (pursuit.hip recompiled with -DMADRL_ABLATE=<n> or extra flags):
#   scripts/variants.sh 1 2 4 8 16      -> scripts/_variants/libmadrl_hip.<n>.so   (git-ignored, travels with gpurun)
# Run one with MADRL_HIP_LIB=scripts/_variants/libmadrl_hip.<n>.so python scripts/sweep_wave.py 4096
set -e
cd "$(dirname "$0")/.."
python -m madrl_amd.build > /dev/null
mkdir -p scripts/_variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math"
for n in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -DMADRL_ABLATE=$n $EXTRA -c madrl_amd/csrc/pursuit.hip -o scripts/_variants/pursuit.$n.o &
done
wait
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/_variants/libmadrl_hip.$n.so scripts/_variants/pursuit.$n.o \
    madrl_amd/csrc/abi.o madrl_amd/csrc/multiwalker.o madrl_amd/csrc/waterworld.o madrl_amd/csrc/wrappers.o
  rm scripts/_variants/pursuit.$n.o
done
ls -la scripts/_variants
