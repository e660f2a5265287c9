cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_sharded_gpu.py tests/test_wrappers_gpu.py tests/test_bench_contract_gpu.py -m gpu -q -x 2>&1 | tail -8
bash scripts/stream_sweep.sh 2>&1 | tee gpurun_out/stream_sweep.txt
