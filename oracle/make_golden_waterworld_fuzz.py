#!/usr/bin/env python
"""Generate tests/golden/waterworld_fuzz_NN.npz: randomly drawn MAWaterWorld configurations run through the UNMODIFIED reference
(/root/reference/madrl_environments/pursuit/waterworld.py) with make_golden_waterworld.run_scenario (same teacher-forcing record).

TEST INFRASTRUCTURE ONLY (build container; outputs are committed).  The files are picked up by every test that replays
tests/golden/waterworld_*.npz: tests/test_oracle_waterworld.py (float64 and float32 C oracle) and tests/test_waterworld_gpu.py (kernel).

Particle counts, n_coop, radii, speeds, sensor count and range, action scale, every reward coefficient, the reward mechanism, the id and
speed-feature switches and the obstacle (fixed at a drawn place, or random per reset) are drawn together from one seed.

    python oracle/make_golden_waterworld_fuzz.py [fuzz_03 ...]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_loader  # noqa: E402
from oracle.make_golden_waterworld import run_scenario  # noqa: E402

N_CASES = 24
MASTER_SEED = 20260925


def draw_case(rng):
    Np = int(rng.randint(1, 9))
    Ne = int(rng.randint(1, 13))
    Npo = int(rng.randint(1, 13))
    kw = dict(n_coop=int(rng.randint(1, min(Np, 3) + 1)), n_poison=Npo,
              radius=float(rng.choice([0.01, 0.015, 0.02, 0.03])), obstacle_radius=float(rng.choice([0.05, 0.1, 0.2, 0.25])),
              ev_speed=float(rng.choice([0.005, 0.01, 0.03, 0.06])), poison_speed=float(rng.choice([0.005, 0.01, 0.04])),
              n_sensors=int(rng.choice([4, 7, 12, 20, 30, 33, 40])), sensor_range=float(rng.choice([0.1, 0.2, 0.35, 0.5])),
              action_scale=float(rng.choice([0.005, 0.01, 0.03, 0.08])), poison_reward=float(rng.choice([-1.0, -0.5, -2.0])),
              food_reward=float(rng.choice([1.0, 10.0, 0.5])), encounter_reward=float(rng.choice([0.05, 0.0, 0.01])),
              control_penalty=float(rng.choice([-0.5, 0.0, -0.1])), reward_mech=str(rng.choice(["local", "global"])),
              addid=bool(rng.rand() < 0.7), speed_features=bool(rng.rand() < 0.7))
    r = rng.rand()
    if r < 0.3:
        kw["obstacle_loc"] = None                                              # drawn by every reset (run_waterworld.py:41)
    elif r < 0.7:
        kw["obstacle_loc"] = np.array([rng.uniform(0.2, 0.8), rng.uniform(0.2, 0.8)])
    run = dict(episodes=int(rng.randint(2, 4)), steps=int(rng.randint(25, 50)), seed=int(rng.randint(1 << 20)),
               action_kind=str(rng.choice(["uniform", "gauss"])), cluster=bool(rng.rand() < 0.7))
    return (Np, Ne), kw, run


def main():
    R = ref_loader.load()
    rng = np.random.RandomState(MASTER_SEED)
    for i in range(N_CASES):
        args, kw, run = draw_case(rng)       # always drawn, so that case i is the same whichever subset is regenerated
        name = "fuzz_%02d" % i
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        run_scenario(R, name, args, kw, **run)


if __name__ == "__main__":
    main()
