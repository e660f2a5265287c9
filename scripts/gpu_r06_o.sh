# round 6: shadow executions at the solver's positions past the register copies (MADRL_MW_SHADOWS=1 variant) against the shipped form, interleaved
# on one box; then the narrow-exec microbenchmarks
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_multiwalker_gpu.py -x -q > gpurun_out/pytest_o.log 2>&1; echo "tests (shipped) rc=$?"; tail -2 gpurun_out/pytest_o.log
for rep in 1 2; do for u in 0 1; do
if [ $u = 0 ]; then unset MADRL_HIP_LIB; else export MADRL_HIP_LIB=$PWD/scripts/_variants/libmadrl_hip.shadows.so; fi
timeout 600 python bench.py --workload multiwalker --steps 50 --warmup 20 --no-cpu-baseline > gpurun_out/bench_w3.log 2>&1; tail -1 gpurun_out/bench_w3.log | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('shadows=$u w3 ms/step %.4f %s one-launch-per-phase %s' % (j['ms_per_step'], j['config'].get('region_ms_per_step'), j['roofline'].get('one_launch_ms')))"
timeout 600 python bench.py --workload multiwalker_w10 --steps 20 --warmup 20 --no-cpu-baseline > gpurun_out/bench_w10.log 2>&1; tail -1 gpurun_out/bench_w10.log | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('shadows=$u w10 ms/step %.4f %s' % (j['ms_per_step'], j['config'].get('region_ms_per_step')))"
done; done
unset MADRL_HIP_LIB
cd scripts/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/exec_passes3 exec_passes3.hip 2>/dev/null && /tmp/exec_passes3
