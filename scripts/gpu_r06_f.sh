# round 6: MultiWalker timing after a solver change -- parity first (kernels vs CPU build, byte for byte), then both BASELINE-line workloads
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_multiwalker_gpu.py -x -q > gpurun_out/pytest_f.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/pytest_f.log
for rep in 1 2; do
timeout 600 python bench.py --workload multiwalker --steps 50 --warmup 20 --no-cpu-baseline > gpurun_out/bench_w3.log 2>&1; tail -1 gpurun_out/bench_w3.log | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('w3 ms/step %.4f %s' % (j['ms_per_step'], j['config'].get('region_ms_per_step')))"
timeout 600 python bench.py --workload multiwalker_w10 --steps 20 --warmup 20 --no-cpu-baseline > gpurun_out/bench_w10.log 2>&1; tail -1 gpurun_out/bench_w10.log | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('w10 ms/step %.4f %s' % (j['ms_per_step'], j['config'].get('region_ms_per_step')))"
done
MW_WINDOWS=30 timeout 600 python scripts/mw_steady.py --one 2>&1 | tail -1
MW_W=10 MW_WINDOWS=30 timeout 600 python scripts/mw_steady.py --one 2>&1 | tail -1
