cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8
for s in 1 2; do python bench.py --workload pursuit_colocate --streams $s --no-cpu-baseline --no-workloads 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('colocate streams $s: %.4f ms/step frac %.3f' % (j['ms_per_step'], r['frac']))"; done
for f in 0 1; do for s in 2 4; do
  if [ $f = 1 ]; then export MADRL_BENCH_MW_FUSED=1; else unset MADRL_BENCH_MW_FUSED; fi
  python bench.py --workload multiwalker --streams $s --no-cpu-baseline --no-workloads 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('multiwalker fused=$f streams $s: %.4f ms/step %s' % (j['ms_per_step'], j['config']['region_ms_per_step']))"
done; done
unset MADRL_BENCH_MW_FUSED
timeout 600 python bench.py > gpurun_out/bench.log 2>gpurun_out/bench.err; echo "bench rc=$?"
tail -1 gpurun_out/bench.log | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('headline %.4g %s  ms/step %.4f  frac %.3f  one-launch frac %.3f' % (j['value'], j['unit'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['one_launch_per_step']['frac']))
for k,v in j.get('workloads',{}).items():
    print(' ', k, ('%.4g ms/step %.4f frac %.3f cpu %.3g' % (v['value'], v['ms_per_step'], v['roofline']['frac'], v['cpu_baseline']['value'])) if 'value' in v else v)
print('cpu_baseline', j.get('cpu_baseline',{}).get('value'))
"
