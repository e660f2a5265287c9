# Policy kernel round: the tests that touch it, the rollout timing table, the rollout workload's profile (kernel stats + PMC traffic)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -k "heuristic or rollout or abi or collector or sharded or bench_contract" > gpurun_out/pytest_policy.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/pytest_policy.log
python scripts/rollout_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/rollout_bench.txt; cat gpurun_out/rollout_bench.txt
bash scripts/profile_workload.sh pursuit_rollout r05_rollout "" > gpurun_out/profile_rollout.log 2>&1; tail -3 gpurun_out/profile_rollout.log | cut -c1-200
cp gpurun_out/rollout_bench.txt gpurun_out/profile/r05_rollout/
