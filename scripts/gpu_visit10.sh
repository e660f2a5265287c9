cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_pursuit_gpu.py tests/test_full_batch_gpu.py tests/test_edge_cases_gpu.py tests/test_curriculum.py tests/test_dropin_callers_gpu.py -m gpu -q -x 2>&1 | tail -8
python bench.py --workload pursuit --no-cpu-baseline --no-workloads 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('pursuit', j['ms_per_step'], j['config']['region_ms_per_step'], j['roofline']['frac'])"
