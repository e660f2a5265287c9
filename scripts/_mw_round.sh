cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_multiwalker_gpu.py tests/test_multiwalker_scenes.py tests/test_multiwalker_envlayer.py -x -q -m gpu > gpurun_out/mw_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/mw_tests.log
for i in 1 2; do
timeout 300 python bench.py --workload multiwalker --no-cpu-baseline --steps 50 > gpurun_out/mw_bench_$i.json 2> gpurun_out/mw_bench_$i.err; echo "bench rc=$?"
python - <<PY
import json
j=json.loads([l for l in open('gpurun_out/mw_bench_$i.json') if l.startswith('{')][-1])
print('mw ms/step', j['ms_per_step'], j['config']['region_ms_per_step'], 'one-launch', j['roofline'].get('one_launch_per_step',{}).get('ms_per_step'))
PY
done
