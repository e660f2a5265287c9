# round 6: the authors' Pursuit shape (30 v 50, obs_range 11): two against four wavefronts per env, interleaved on ONE box (boxes differ by +-5 %)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2 3; do for u in 4 2; do
if [ $u = 4 ]; then unset MADRL_HIP_LIB; else export MADRL_HIP_LIB=$PWD/scripts/_variants/libmadrl_hip.pursuit.nw2.so; fi
timeout 600 python bench.py --workload pursuit_authors --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/bench_pa.log 2>&1; tail -1 gpurun_out/bench_pa.log | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('authors NW=$u ms/step %.4f %s frac %.3f one %s %s' % (j['ms_per_step'], j['config']['region_ms_per_step'], j['roofline']['frac'], j['roofline'].get('one_launch_ms'), j['roofline']['kernel'][:60]))"
done; done
