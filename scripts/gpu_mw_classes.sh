cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_multiwalker_gpu.py tests/test_multiwalker_envlayer.py tests/test_multiwalker_scenes.py -m gpu -q > gpurun_out/mw_tests.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/mw_tests.log
