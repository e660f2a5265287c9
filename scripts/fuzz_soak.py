"""One-off soak of the free-running drawn-configuration tests beyond what tests/ runs every time: many more drawn Pursuit / Waterworld /
hostage configurations, each stepped by the kernels and by the C oracle side by side (nothing injected: every draw from the Philox
contract on both sides), every output compared bit for bit.  Usage (GPU box): python scripts/fuzz_soak.py [n_pursuit n_waterworld n_hostage]"""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_pursuit_gpu as tp, test_waterworld_gpu as tw, test_hostage_gpu as th

n_p, n_w, n_h = (int(a) for a in (sys.argv[1:4] + ["120", "80", "60"])[:3])
bad = []
t0 = time.time()
kinds = {}
for seed in (1, 2, 3):
    for i, (maps, cfg) in enumerate(tp._drawn_cases(n=n_p // 3, seed=20270000 + seed)):
        big = cfg["n_pursuers"] + cfg["n_evaders"] > 64
        for kernel in ("auto", "generic"):
            try:
                tp._free_run(maps, dict(cfg), kernel, 96 if big else 256, 50, expect_catches=False)
            except Exception as e:  # noqa
                bad.append(("pursuit", seed, i, kernel, repr(e)[:300])); traceback.print_exc()
print("pursuit: %d drawn configurations x kernels, %d failures, %.0f s" % (n_p // 3 * 3, len(bad), time.time() - t0), flush=True)
t0 = time.time(); nb = len(bad)
for i in range(10, 10 + n_w):
    tw.CASES["drawn_%d" % i] = tw._drawn_case(i)
    try:
        tw._vs_f32_oracle("drawn_%d" % i, teacher_forced=False)
    except AssertionError as e:
        if str(e) != "no catches":
            bad.append(("waterworld", i, repr(e)[:300])); traceback.print_exc()
    except Exception as e:  # noqa
        bad.append(("waterworld", i, repr(e)[:300])); traceback.print_exc()
print("waterworld: %d drawn configurations, %d failures, %.0f s" % (n_w, len(bad) - nb, time.time() - t0), flush=True)
t0 = time.time(); nb = len(bad)
for i in range(8, 8 + n_h):
    try:
        th.test_drawn_configurations_free_running_vs_f32_oracle(i)
    except AssertionError as e:
        if "n_done" in str(e) or "n_resp" in str(e):
            continue
        bad.append(("hostage", i, repr(e)[:300])); traceback.print_exc()
    except Exception as e:  # noqa
        bad.append(("hostage", i, repr(e)[:300])); traceback.print_exc()
print("hostage: %d drawn configurations, %d failures, %.0f s" % (n_h, len(bad) - nb, time.time() - t0), flush=True)
for b in bad:
    print("FAIL", b)
print("fuzz soak:", "clean" if not bad else "%d FAILURES" % len(bad))
