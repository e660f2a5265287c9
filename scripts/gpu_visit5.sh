cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
show() { tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; o=r.get('one_launch_per_step',{})
print('$1: %.4f ms/step %s | one launch %.4f' % (j['ms_per_step'], j['config']['region_ms_per_step'], o.get('ms_per_step',0)))
for k,v in j.get('workloads',{}).items(): print('   ', k, v['ms_per_step'], v['config']['region_ms_per_step'], v['roofline'].get('one_launch_per_step',{}).get('ms_per_step'))"; }
python bench.py --workload waterworld --steps 200 --warmup 20 --no-workloads --no-cpu-baseline 2>/dev/null | show "ww K=200 no cpu"
python bench.py --workload waterworld --steps 200 --warmup 20 --no-workloads 2>/dev/null | show "ww K=200 with cpu baseline (after)"
python bench.py --no-cpu-baseline 2>/dev/null | show "default, no cpu baselines"
MADRL_BENCH_CPU_BUDGET=0.2 python bench.py 2>/dev/null | show "default, short cpu baselines"
for s in 1; do MADRL_BENCH_MW_FUSED=1 python bench.py --workload multiwalker --streams $s --no-cpu-baseline --no-workloads 2>/dev/null | show "mw fused S=$s"; done
