# round 6: Waterworld / hostage sensing one pass at a time -- parity, timing (several runs: the regions are noisy), instruction mix
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_waterworld_gpu.py tests/test_hostage_gpu.py tests/test_full_batch_gpu.py tests/test_wrappers_gpu.py tests/test_round6_regressions_gpu.py -x -q -k "waterworld or hostage or view or standard" > gpurun_out/pytest_d.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/pytest_d.log
for rep in 1 2 3; do for w in waterworld hostage; do
timeout 600 python bench.py --workload $w --steps 300 --warmup 50 --no-cpu-baseline > gpurun_out/bench_$w.log 2>&1; tail -1 gpurun_out/bench_$w.log | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$w ms/step %.4f %s frac %.3f one %s' % (j['ms_per_step'], j['config']['region_ms_per_step'], j['roofline']['frac'], j['roofline'].get('one_launch_ms')))"
done; done
bash scripts/pmc_mix.sh waterworld waterworld_kernel 16384 > gpurun_out/pmc_mix_ww.log 2>&1; grep "SQ_INSTS\|SQ_WAVE_CYCLES\|SQ_WAIT\|SQ_ACTIVE_INST_VALU\|SQ_ACTIVE_INST_SCA\|SQ_INST_CYCLES_SALU\|GRBM" gpurun_out/pmc_mix_ww.log
bash scripts/pmc_mix.sh hostage hostage_kernel 16384 > gpurun_out/pmc_mix_hw.log 2>&1; grep "SQ_INSTS_VALU\|SQ_INSTS_SALU\|SQ_INSTS_LDS" gpurun_out/pmc_mix_hw.log
