# round 6: the env's record found again from the lane id after the solver in EVERY class (four-lane one-launch kernel: 507 -> 417 VGPRs), and with the
# registers that frees a fourth manifold per lane in registers (469), against the previous commit's kernels; interleaved on one box
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
MADRL_HIP_LIB=$PWD/scripts/_variants/libmadrl_hip.m4.so timeout 1500 python -m pytest tests/test_multiwalker_gpu.py -x -q -k "bit_for_bit or auto_reset" 2>&1 | tail -1
timeout 1500 python -m pytest tests/test_multiwalker_gpu.py -x -q 2>&1 | tail -1
for rep in 1 2 3; do for u in again again_mreg4 head; do
if [ $u = again ]; then unset MADRL_HIP_LIB; elif [ $u = head ]; then export MADRL_HIP_LIB=$PWD/scripts/_variants/libmadrl_hip.head.so; else export MADRL_HIP_LIB=$PWD/scripts/_variants/libmadrl_hip.m4.so; fi
timeout 600 python bench.py --workload multiwalker --steps 50 --warmup 20 --no-cpu-baseline > gpurun_out/bench_w3.log 2>&1; tail -1 gpurun_out/bench_w3.log | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$u w3 ms/step %.4f %s one-launch-per-phase %s' % (j['ms_per_step'], j['config'].get('region_ms_per_step'), j['roofline'].get('one_launch_ms')))"
done; done
for u in again head; do
if [ $u = again ]; then unset MADRL_HIP_LIB; else export MADRL_HIP_LIB=$PWD/scripts/_variants/libmadrl_hip.head.so; fi
timeout 600 python bench.py --workload multiwalker_w10 --steps 20 --warmup 20 --no-cpu-baseline > gpurun_out/bench_w10.log 2>&1; tail -1 gpurun_out/bench_w10.log | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$u w10 ms/step %.4f %s' % (j['ms_per_step'], j['config'].get('region_ms_per_step')))"
done
