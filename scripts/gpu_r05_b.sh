cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python scripts/rollout_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/rollout_bench.txt
(cd /tmp && ROLLOUT_ONLY=single timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_rollout -o rollout -- python $GRAFT_REPO_ROOT/scripts/rollout_bench.py > $GRAFT_REPO_ROOT/gpurun_out/prof_rollout.log 2>&1)
find gpurun_out/prof_rollout -name '*kernel_stats.csv' | head -1 | xargs head -8 | cut -c1-220
bash scripts/gpu_round.sh
