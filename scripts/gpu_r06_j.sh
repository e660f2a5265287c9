# round 6: the authors' Pursuit shape (30 v 50, obs_range 11), four wavefronts per env: how many slots of the rolled loop in flight together
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
for u in 0 1 4 8; do
if [ $u = 0 ]; then unset MADRL_HIP_LIB; else export MADRL_HIP_LIB=$PWD/scripts/_variants/libmadrl_hip.pursuit.$u.so; fi
timeout 600 python bench.py --workload pursuit_authors --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/bench_pa.log 2>&1; tail -1 gpurun_out/bench_pa.log | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('authors unroll $u (0 = shipped, 2) ms/step %.4f %s frac %.3f one %s' % (j['ms_per_step'], j['config']['region_ms_per_step'], j['roofline']['frac'], j['roofline'].get('one_launch_ms')))"
done
