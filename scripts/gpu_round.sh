# One GPU call that produces what a round is judged by: smoke, the whole GPU test suite, the bench line as the driver asks for it
# (python bench.py --gpus 1 --steps 20 --warmup 5) + its long record.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"
tail -2 gpurun_out/smoke.log
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -8 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench.log 2>gpurun_out/bench.err; echo "bench rc=$?"
tail -1 gpurun_out/bench.log | wc -c
tail -1 gpurun_out/bench.log | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('headline %.4g %s  ms/step %.4f %s frac %.3f (kernel %.3f)  one-launch %.4f' % (j['value'], j['unit'], j['ms_per_step'], j['config']['region_ms_per_step'], j['roofline']['frac'], j['roofline']['frac_kernel'], j['roofline'].get('one_launch_ms', 0)))
for k,v in j.get('workloads',{}).items():
    print(' ', k, ('%.4g ms/step %.4f frac %.3f one-launch %s cpu %s' % (v['value'], v['ms_per_step'], v['roofline']['frac'], v['roofline'].get('one_launch_ms'), v.get('cpu_baseline',{}).get('value'))) if 'value' in v else v)
print('cpu_baseline', j.get('cpu_baseline',{}).get('value'))
"
