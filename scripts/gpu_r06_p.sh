# round 6: "record found again from the lane id" in every class (fewer registers: 507 -> 422 in the four-lane one-launch kernel) against the
# previous commit's kernels, interleaved on one box
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2 3; do for u in new head; do
if [ $u = new ]; then unset MADRL_HIP_LIB; else export MADRL_HIP_LIB=$PWD/scripts/_variants/libmadrl_hip.head.so; fi
timeout 600 python bench.py --workload multiwalker --steps 50 --warmup 20 --no-cpu-baseline > gpurun_out/bench_w3.log 2>&1; tail -1 gpurun_out/bench_w3.log | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$u w3 ms/step %.4f %s one-launch-per-phase %s' % (j['ms_per_step'], j['config'].get('region_ms_per_step'), j['roofline'].get('one_launch_ms')))"
timeout 600 python bench.py --workload multiwalker_w10 --steps 20 --warmup 20 --no-cpu-baseline > gpurun_out/bench_w10.log 2>&1; tail -1 gpurun_out/bench_w10.log | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$u w10 ms/step %.4f %s' % (j['ms_per_step'], j['config'].get('region_ms_per_step')))"
done; done
