cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "=== stats stream ubench"; timeout 120 scripts/ubench/stats_stream 2>&1 | tee gpurun_out/stats_stream.txt
echo "=== pursuit split"; timeout 300 python scripts/pursuit_split.py 2>&1 | tee gpurun_out/pursuit_split.txt
echo "=== ww reset tail"; timeout 300 python scripts/ww_reset_tail.py 2>&1 | tee gpurun_out/ww_reset_tail.txt
echo "=== mw tests + bench"; timeout 600 python -m pytest tests/test_multiwalker_gpu.py tests/test_multiwalker_scenes.py tests/test_advice_regressions.py -m gpu -q -x 2>&1 | tail -5
timeout 300 python bench.py --workload multiwalker --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('multiwalker ms/step', j['ms_per_step'], j['config']['region_ms_per_step'])"
MADRL_HIP_LIB=$PWD/scripts/_variants/libmadrl_hip.multiwalker.1.so timeout 300 python scripts/mw_timing.py > gpurun_out/mw_timing2.txt 2>&1; grep -A12 "^step 2" gpurun_out/mw_timing2.txt | head -30
