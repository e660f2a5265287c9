cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
bash scripts/gpu_round.sh 2>&1 | tail -30
echo "=== mw timing"
MADRL_HIP_LIB=$PWD/scripts/_variants/libmadrl_hip.multiwalker.1.so timeout 300 python scripts/mw_timing.py > gpurun_out/mw_timing.txt 2>&1; echo "rc=$?"
tail -60 gpurun_out/mw_timing.txt
