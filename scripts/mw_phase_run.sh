#!/bin/bash
# kernel times of the MultiWalker step in steady state for the current build and environment
#   scripts/mw_phase_run.sh <tag>      (inside gpurun)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/mwprof_$1
rocprofv3 --kernel-trace -d gpurun_out/mwprof_$1 -o mw -- python scripts/mw_steady.py --quick --one > /dev/null 2>&1
echo "== $1"; python scripts/mw_kernels.py gpurun_out/mwprof_$1/mw_results.db; rm -rf gpurun_out/mwprof_$1
