# round 6: the headline kernel at 4 / 5 (shipped) / 6 resident wavefronts per SIMD, interleaved on one box (re-check of rounds 2-4's tuning after ABI 7)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2; do for u in 5 4 6; do
if [ $u = 5 ]; then unset MADRL_HIP_LIB; else export MADRL_HIP_LIB=$PWD/scripts/_variants/libmadrl_hip.pursuit.$u.so; fi
timeout 600 python bench.py --workload pursuit --steps 300 --warmup 50 --no-cpu-baseline --no-workloads > gpurun_out/bench_p.log 2>&1; tail -1 gpurun_out/bench_p.log | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('pursuit waves/SIMD=$u ms/step %.4f %s frac %.3f one %s' % (j['ms_per_step'], j['config']['region_ms_per_step'], j['roofline']['frac'], j['roofline'].get('one_launch_ms')))"
done; done
