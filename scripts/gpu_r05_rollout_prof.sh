cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
cd /tmp && ROLLOUT_ONLY=single timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_rollout -o rollout -- python $GRAFT_REPO_ROOT/scripts/rollout_bench.py > $GRAFT_REPO_ROOT/gpurun_out/prof_rollout.log 2>&1
cd $GRAFT_REPO_ROOT; tail -3 gpurun_out/prof_rollout.log; find gpurun_out/prof_rollout -name '*kernel_stats.csv' | head -1 | xargs head -12 | cut -c1-200
