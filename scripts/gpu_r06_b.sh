# round 6, second GPU call: the long-row (LDS slot table) Pursuit kernel on the authors' shapes
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_pursuit_gpu.py tests/test_full_batch_gpu.py tests/test_round6_regressions_gpu.py tests/test_sharded_gpu.py -x -q -k "authors or c5 or group or round6 or sub_batch or long_rollout or full_batch or flags or view" > gpurun_out/pytest_b.log 2>&1; echo "tests rc=$?"; tail -12 gpurun_out/pytest_b.log
timeout 600 python scripts/authors_shape.py 16384 50 > gpurun_out/authors_shape.log 2>&1; tail -6 gpurun_out/authors_shape.log
for w in pursuit_authors pursuit_c5 hostage; do
timeout 600 python bench.py --workload $w --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/bench_$w.log 2>&1; tail -1 gpurun_out/bench_$w.log | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$w ms/step %.4f %s frac %.3f one %s kernel %s' % (j['ms_per_step'], j['config']['region_ms_per_step'], j['roofline']['frac'], j['roofline'].get('one_launch_ms'), j['roofline']['kernel']))"
done
