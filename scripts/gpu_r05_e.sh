cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
echo "--- pursuit headline kernel, one launch per step, mask equilibrium: shipped | every float4 full | every float4 full + 20 more vector instructions per slot"
for v in "" "$PWD/scripts/_variants/libmadrl_hip.pursuit.2.so" "$PWD/scripts/_variants/libmadrl_hip.pursuit.34.so" "" "$PWD/scripts/_variants/libmadrl_hip.pursuit.34.so"; do echo "lib=${v:-shipped}"; MADRL_HIP_LIB=$v WINDOWS=6 timeout 300 python scripts/zmask_drift.py 2>&1 | grep -v amdgpu.ids | tail -1; done | tee gpurun_out/wave_full_store_bound.txt
echo "--- rollout (policy kernel: 16 lanes per row, persistent)"; timeout 600 python scripts/rollout_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/rollout_bench.txt
