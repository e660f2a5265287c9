# round 6: the one-launch kernel of the sixteen-lane MultiWalker class (record pointers found again from the lane id after the sweeps):
# parity against the CPU build, then ten walkers timed as one launch and as three
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_multiwalker_gpu.py -x -q > gpurun_out/pytest_e.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/pytest_e.log
for f in 1 0; do
MADRL_BENCH_MW_FUSED=$f timeout 600 python bench.py --workload multiwalker_w10 --steps 20 --warmup 20 --no-cpu-baseline > gpurun_out/bench_w10_f$f.log 2>&1; tail -1 gpurun_out/bench_w10_f$f.log | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('w10 fused=$f ms/step %.4f %s' % (j['ms_per_step'], j['config'].get('region_ms_per_step')))"
done
for f in 1 0; do
MADRL_BENCH_MW_FUSED=$f timeout 600 python bench.py --workload multiwalker --steps 50 --warmup 20 --no-cpu-baseline > gpurun_out/bench_w3_f$f.log 2>&1; tail -1 gpurun_out/bench_w3_f$f.log | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('w3 fused=$f ms/step %.4f %s' % (j['ms_per_step'], j['config'].get('region_ms_per_step')))"
done
