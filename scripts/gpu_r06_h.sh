# round 6: ten walkers, the solver launch at two wavefronts per SIMD (three launches per step) against the one-launch kernel
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_multiwalker_gpu.py -x -q -k "9 or 10" > gpurun_out/pytest_h.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/pytest_h.log
for rep in 1 2; do for f in 0 1; do
MADRL_BENCH_MW_FUSED=$f timeout 600 python bench.py --workload multiwalker_w10 --steps 20 --warmup 20 --no-cpu-baseline > gpurun_out/bench_w10_f$f.log 2>&1; tail -1 gpurun_out/bench_w10_f$f.log | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('w10 fused=$f ms/step %.4f %s' % (j['ms_per_step'], j['config'].get('region_ms_per_step')))"
done; done
for s in 1 2 8; do
MADRL_BENCH_MW_FUSED=0 timeout 600 python bench.py --workload multiwalker_w10 --steps 20 --warmup 20 --no-cpu-baseline --streams $s > gpurun_out/bench_w10_s$s.log 2>&1; tail -1 gpurun_out/bench_w10_s$s.log | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('w10 three launches streams=$s ms/step %.4f %s' % (j['ms_per_step'], j['config'].get('region_ms_per_step')))"
done
