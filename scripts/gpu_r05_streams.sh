cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python scripts/pursuit_split.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/pursuit_split.txt
timeout 600 python scripts/rollout_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/rollout_bench.txt
