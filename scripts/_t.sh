cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
MADRL_HIP_LIB=$PWD/scripts/_variants/libmadrl_hip.multiwalker.1.so timeout 300 python scripts/mw_timing.py > gpurun_out/mw_timing.txt 2>&1; echo rc=$?
tail -30 gpurun_out/mw_timing.txt
