# round 6, first GPU call: the new regression tests, the whole GPU suite, the driver-style bench line, the authors' shapes on the generic kernel
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_round6_regressions_gpu.py -x -q > gpurun_out/pytest_r6.log 2>&1; echo "r6 tests rc=$?"; tail -15 gpurun_out/pytest_r6.log
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench.log 2>gpurun_out/bench.err; echo "bench rc=$?"; tail -3 gpurun_out/bench.err
tail -1 gpurun_out/bench.log | wc -c
tail -1 gpurun_out/bench.log | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('headline %.4g ms/step %.4f %s frac %.3f one-launch %.4f api %s api-one %s' % (j['value'], j['ms_per_step'], j['config']['region_ms_per_step'], j['roofline']['frac'], j['roofline'].get('one_launch_ms', 0), j.get('python_api_ms_per_step'), j.get('python_api_one_launch_ms')))
for k,v in j.get('workloads',{}).items():
    print(' ', k, ('%.4g ms/step %.4f frac %.3f one-launch %s api %s' % (v['value'], v['ms_per_step'], v['roofline']['frac'], v['roofline'].get('one_launch_ms'), v.get('python_api_ms'))) if 'value' in v else v)
"
timeout 600 python bench.py --steps 500 --warmup 50 --no-workloads > gpurun_out/bench500.log 2>&1; tail -1 gpurun_out/bench500.log | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('K=500 headline ms/step %.4f api %s one %.4f api-one %s' % (j['ms_per_step'], j.get('python_api_ms_per_step'), j['roofline'].get('one_launch_ms',0), j.get('python_api_one_launch_ms')))"
timeout 600 python scripts/authors_shape.py 16384 50 > gpurun_out/authors_shape.log 2>&1; cat gpurun_out/authors_shape.log | tail -6
