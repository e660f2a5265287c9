# every bench workload as S sub-batches per GPU on their own HIP streams (bench.py sets GPU_MAX_HW_QUEUES=8)
cd $GRAFT_REPO_ROOT 2>/dev/null || cd "$(dirname "$0")/.."
for w in ${WORKLOADS:-pursuit pursuit_c5 pursuit_colocate waterworld hostage waterworld_std multiwalker}; do
  for s in ${STREAMS:-1 2 4 8}; do
    python bench.py --workload $w --streams $s --no-cpu-baseline --no-workloads 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']
print('%-18s streams %d: %.4f ms/step (regions %s) frac %.3f' % ('$w', $s, j['ms_per_step'], j['config']['region_ms_per_step'], r['frac']))"
  done
done
