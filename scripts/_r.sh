cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_pursuit_gpu.py tests/test_multiwalker_envlayer.py -m gpu -q -k "drawn or every_lane or free_running" > gpurun_out/pytest_new.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/pytest_new.log
