#!/bin/bash
# Build profiling variants of the library: one source recompiled with -D<MACRO>=<n> (never the shipped library).
#   scripts/variants.sh 1 2 4 8 16                         -> pursuit.hip with -DMADRL_ABLATE=<n>
#   SRC=waterworld MACRO=MADRL_WW_ABLATE scripts/variants.sh 1 2
# -> scripts/_variants/libmadrl_hip.<src>.<n>.so (git-ignored, travels with gpurun); run one with
#   MADRL_HIP_LIB=$PWD/scripts/_variants/libmadrl_hip.pursuit.2.so python scripts/sweep_wave.py 4096
set -e
cd "$(dirname "$0")/.."
SRC=${SRC:-pursuit}; MACRO=${MACRO:-MADRL_ABLATE}
python -m madrl_amd.build > /dev/null
mkdir -p scripts/_variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math"
for n in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -D$MACRO=$n $EXTRA -c madrl_amd/csrc/$SRC.hip -o scripts/_variants/$SRC.$n.o &
done
wait
for n in "$@"; do
  OTHERS=$(ls madrl_amd/csrc/*.o | grep -v "/$SRC.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/_variants/libmadrl_hip.$SRC.$n.so scripts/_variants/$SRC.$n.o $OTHERS
  rm scripts/_variants/$SRC.$n.o
done
ls scripts/_variants
