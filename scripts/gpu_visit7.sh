cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
show() { tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; o=r.get('one_launch_per_step',{})
print('$1: %.4f ms/step %s | one launch %.4f' % (j['ms_per_step'], j['config']['region_ms_per_step'], o.get('ms_per_step',0)))
for k,v in j.get('workloads',{}).items(): print('   ', k, v['ms_per_step'], v['config']['region_ms_per_step'], v['roofline'].get('one_launch_per_step',{}).get('ms_per_step'))"; }
python bench.py --no-cpu-baseline 2>/dev/null | show "default (GPU_MAX_HW_QUEUES=8, eager streams)"
GPU_MAX_HW_QUEUES=4 python bench.py --no-cpu-baseline 2>/dev/null | show "default with 4 hw queues"
timeout 300 python -m pytest tests/test_bench_contract_gpu.py -m gpu -q 2>&1 | tail -3
