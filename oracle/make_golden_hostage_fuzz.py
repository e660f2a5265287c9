#!/usr/bin/env python
"""Generate tests/golden/hostage_fuzz_NN.npz: randomly drawn ContinuousHostageWorld configurations run through the UNMODIFIED
reference (/root/reference/madrl_environments/hostage.py) with make_golden_hostage.run (same teacher-forcing record).

TEST INFRASTRUCTURE ONLY (build container; outputs are committed).  The files are picked up by every test that replays
tests/golden/hostage_*.npz (tests/test_oracle_hostage.py, tests/test_hostage_gpu.py).

    python oracle/make_golden_hostage_fuzz.py [fuzz_03 ...]
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_loader  # noqa: E402
from oracle.make_golden_hostage import run  # noqa: E402

N_CASES = 16
MASTER_SEED = 20260926


def draw_case(rng):
    n_good = int(rng.randint(1, 6))
    n_host = int(rng.randint(1, 12))
    n_bad = int(rng.randint(1, 8))
    args = (n_good, n_host, n_bad, int(rng.randint(1, min(n_good, 3) + 1)), int(rng.randint(1, min(n_good, 2) + 1)))
    kw = dict(radius=float(rng.choice([0.01, 0.015, 0.025])), bad_speed=float(rng.choice([0.005, 0.01, 0.03])),
              n_sensors=int(rng.choice([6, 12, 20, 30, 36])), sensor_range=float(rng.choice([0.15, 0.2, 0.3, 0.45])),
              action_scale=float(rng.choice([0.01, 0.02, 0.04])), save_reward=float(rng.choice([5.0, 1.0])),
              hit_reward=float(rng.choice([-1.0, -0.25])), encounter_reward=float(rng.choice([0.01, 0.0, 0.1])),
              not_saved_reward=float(rng.choice([-3.0, -1.0])), bomb_reward=float(rng.choice([-5.0, -2.0])),
              bomb_radius=float(rng.choice([0.05, 0.08])), key_radius=float(rng.choice([0.0075, 0.02])),
              control_penalty=float(rng.choice([-0.1, 0.0, -0.5])), reward_mech=str(rng.choice(["global", "local"])),
              addid=bool(rng.rand() < 0.7))
    run_kw = dict(episodes=int(rng.randint(2, 4)), steps=int(rng.randint(30, 70)), seed=int(rng.randint(1 << 20)), herd=bool(rng.rand() < 0.75))
    return args, kw, run_kw


def main():
    ref_loader.load()
    H = importlib.import_module("madrl_environments.hostage")
    rng = np.random.RandomState(MASTER_SEED)
    for i in range(N_CASES):
        args, kw, run_kw = draw_case(rng)    # always drawn, so that case i is the same whichever subset is regenerated
        name = "fuzz_%02d" % i
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        run(H, name, args, kw, **run_kw)


if __name__ == "__main__":
    main()
