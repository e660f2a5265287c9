cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_multiwalker_gpu.py tests/test_multiwalker_scenes.py tests/test_advice_regressions.py -m gpu -q -x 2>&1 | tail -4
for s in 1 4; do python bench.py --workload multiwalker --streams $s --no-cpu-baseline --no-workloads 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('multiwalker streams $s: %.4f ms/step %s' % (j['ms_per_step'], j['config']['region_ms_per_step']))"; done
MADRL_HIP_LIB=$PWD/scripts/_variants/libmadrl_hip.multiwalker.1.so timeout 300 python scripts/mw_timing.py > gpurun_out/mw_timing3.txt 2>&1; grep -A10 "^step 2, continuous" gpurun_out/mw_timing3.txt | head -14; grep "^step 2, solve" gpurun_out/mw_timing3.txt
