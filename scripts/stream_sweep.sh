# every bench workload as 1 / 2 / 4 sub-batches per GPU on their own HIP streams
cd $GRAFT_REPO_ROOT 2>/dev/null || cd "$(dirname "$0")/.."
for w in pursuit pursuit_c5 pursuit_colocate waterworld hostage; do
  for s in 2 4; do
    python bench.py --workload $w --streams $s --no-cpu-baseline --no-workloads 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; o=r.get('one_launch_per_step',{})
print('%-18s streams %d: %.4f ms/step (regions %s) frac %.3f | one launch per step: %.4f ms frac %.3f' % ('$w', $s, j['ms_per_step'], j['config']['region_ms_per_step'], r['frac'], o.get('ms_per_step',0), o.get('frac',0)))"
  done
done
for f in auto 0; do for s in 1 2 4; do
    python bench.py --workload waterworld_std --std-fused $f --streams $s --no-cpu-baseline --no-workloads 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']
print('waterworld_std fused=$f streams $s: %.4f ms/step frac %.3f (%d B/env)' % (j['ms_per_step'], r['frac'], r['algorithmic_bytes_per_env_step']))"
done; done
for s in 1 2 4 8; do
    python bench.py --workload multiwalker --streams $s --no-cpu-baseline --no-workloads 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('multiwalker streams $s: %.4f ms/step %s' % (j['ms_per_step'], j['config']['region_ms_per_step']))"
done
