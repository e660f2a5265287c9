cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python scripts/rollout_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/rollout_bench.txt
(cd /tmp && ROLLOUT_ONLY=single timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_rollout -o rollout -- python $GRAFT_REPO_ROOT/scripts/rollout_bench.py > $GRAFT_REPO_ROOT/gpurun_out/prof_rollout.log 2>&1)
find gpurun_out/prof_rollout -name '*kernel_stats.csv' | head -1 | xargs head -4 | cut -c1-260
timeout 900 python -m pytest tests/test_pursuit_gpu.py tests/test_heuristics_gpu.py tests/test_edge_cases_gpu.py tests/test_sharded_gpu.py tests/test_rollout.py tests/test_dropin_callers_gpu.py -m gpu -q 2>&1 | tail -5
