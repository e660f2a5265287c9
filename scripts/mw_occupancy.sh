#!/bin/bash
# MultiWalker, three walkers: 16 envs per wavefront (the four-lane class it runs on) against 8 and 4 envs per wavefront (the same walkers on
# the eight- and sixteen-lane classes: fewer envs in lockstep per wavefront, twice / four times the wavefronts), at 16 384 and 32 768 envs.
# Builds a measurement library whose C ABI honours MADRL_MW_MIN_CLASS (never the shipped one), then times scripts/mw_steady.py.
#   scripts/mw_occupancy.sh build     (here, no GPU)        scripts/mw_occupancy.sh run   (on the GPU box)
cd "$(dirname "$0")/.."
if [ "$1" = "build" ]; then
  python -m madrl_amd.build > /dev/null
  mkdir -p scripts/_variants
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DMADRL_EXPERIMENTS -c madrl_amd/csrc/multiwalker.hip -o scripts/_variants/mw_dispatch_x.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/_variants/libmadrl_hip.mwclass.so scripts/_variants/mw_dispatch_x.o $(ls madrl_amd/csrc/*.o | grep -v '/multiwalker.o')
  rm scripts/_variants/mw_dispatch_x.o; ls -la scripts/_variants/libmadrl_hip.mwclass.so
else
  for n in 16384 32768; do for c in 4 8 10; do
    echo "envs $n, three walkers on the class for $c walkers ($((64 / (c == 10 ? 16 : c))) envs per wavefront):"
    MADRL_HIP_LIB=$PWD/scripts/_variants/libmadrl_hip.mwclass.so MADRL_MW_MIN_CLASS=$c MW_N=$n MW_W=3 MW_WINDOWS=10 timeout 300 python scripts/mw_steady.py --one 2>&1 | tail -1
  done; done
fi
