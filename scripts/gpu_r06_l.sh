# round 6: three walkers, one-launch steps: how many sub-batches (a launch ends with its slowest wavefront: fewer wavefronts per launch, shorter tails)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
for s in 4 8 16 32 2; do
GPU_MAX_HW_QUEUES=${Q:-8} MADRL_BENCH_MW_FUSED=1 timeout 600 python bench.py --workload multiwalker --steps 50 --warmup 20 --no-cpu-baseline --streams $s > gpurun_out/bench_w3_s$s.log 2>&1; tail -1 gpurun_out/bench_w3_s$s.log | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('w3 one-launch steps, streams=$s ms/step %.4f %s' % (j['ms_per_step'], j['config'].get('region_ms_per_step')))"
done
