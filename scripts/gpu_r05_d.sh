cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
echo "--- rollout (policy kernel v4)"; timeout 600 python scripts/rollout_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/rollout_bench.txt
echo "--- multiwalker W=3 on larger classes"; bash scripts/mw_occupancy.sh run 2>&1 | grep -v amdgpu.ids | tee gpurun_out/mw_occupancy.txt
echo "--- pursuit headline kernel: shipped vs every float4 full (the bound of any stale-value scheme)"
for v in "" "$PWD/scripts/_variants/libmadrl_hip.pursuit.2.so"; do echo "lib=${v:-shipped}"; MADRL_HIP_LIB=$v WINDOWS=7 timeout 300 python scripts/zmask_drift.py 2>&1 | grep -v amdgpu.ids | tail -2; done | tee gpurun_out/wave_full_store_bound.txt
