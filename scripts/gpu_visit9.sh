cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_sharded_gpu.py tests/test_rollout.py -m gpu -q 2>&1 | tail -4
timeout 600 python scripts/rollout_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/rollout_bench.txt
