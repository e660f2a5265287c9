# round 6: Waterworld sensing one pass at a time -- parity + timing
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_waterworld_gpu.py tests/test_full_batch_gpu.py tests/test_wrappers_gpu.py tests/test_sharded_gpu.py tests/test_round6_regressions_gpu.py -x -q -k "waterworld or view or standard" > gpurun_out/pytest_c.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/pytest_c.log
for w in waterworld hostage; do
timeout 600 python bench.py --workload $w --steps 300 --warmup 30 --no-cpu-baseline > gpurun_out/bench_$w.log 2>&1; tail -1 gpurun_out/bench_$w.log | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$w ms/step %.4f %s frac %.3f one %s' % (j['ms_per_step'], j['config']['region_ms_per_step'], j['roofline']['frac'], j['roofline'].get('one_launch_ms')))"
done
