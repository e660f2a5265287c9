# the MultiWalker GPU tests of all three capacity classes, then steady-state step times for 3, 5, 8, 10 walkers (scripts/mw_steady.py)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_multiwalker_gpu.py tests/test_multiwalker_envlayer.py tests/test_multiwalker_scenes.py tests/test_edge_cases_gpu.py tests/test_advice_regressions.py -m gpu -q > gpurun_out/mw_tests.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/mw_tests.log
for w in 3 5 8 10; do echo "n_walkers $w:"; MW_W=$w MW_WINDOWS=8 timeout 300 python scripts/mw_steady.py --one 2>&1 | tail -1; done | tee gpurun_out/mw_steady_classes.txt
