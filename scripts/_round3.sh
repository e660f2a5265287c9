cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_multiwalker_envlayer.py -m gpu -q > gpurun_out/pytest_envlayer.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/pytest_envlayer.log
