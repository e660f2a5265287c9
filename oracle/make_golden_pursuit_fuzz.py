#!/usr/bin/env python
"""Generate tests/golden/pursuit_fuzz_NN.npz: randomly drawn PursuitEvade configurations run through the UNMODIFIED reference
(/root/reference/madrl_environments/pursuit/pursuit_evade.py) with make_golden_pursuit.run_scenario -- same recording, same pinning of
the reference's randomness (injected positions, scripted evader moves).

TEST INFRASTRUCTURE ONLY.  Runs in the build container (the reference tree is not present on the GPU box); its outputs are committed
and are picked up by the tests that replay every tests/golden/pursuit_*.npz (tests/test_oracle_pursuit.py through the C oracle,
tests/test_pursuit_gpu.py through the kernels).

The hand-written scenarios of make_golden_pursuit.py each aim at one quirk.  These aim at none: map size and shape, building
density, walls on the border rows (need_to_surround's skipped neighbours, pursuit_evade.py:536), agent counts from 1 to a few dozen,
odd and even obs_range up to windows wider than the map, both catch modes with n_catch 1..4, both reward mechanisms with random
coefficients, flat and (R, R, 4) observations with and without the id, map pools, random_opponents, agents created inside buildings
are all drawn together from one seed, so that combinations nobody thought of get replayed too.

    python oracle/make_golden_pursuit_fuzz.py            # all of them
    python oracle/make_golden_pursuit_fuzz.py fuzz_07     # only the named ones
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_loader  # noqa: E402
from oracle.make_golden_pursuit import run_scenario  # noqa: E402

N_CASES = 48
MASTER_SEED = 20260924


def random_map(rng, xs, ys):
    """0 = free, -1 = building (TwoDMaps conventions); at least a third of the cells stay free"""
    kind = rng.randint(4)
    m = np.zeros((xs, ys), np.int32)
    if kind == 0:
        return m                                           # open field
    if kind == 1:                                          # a central block like rectangle_map, random extent
        x0, x1 = sorted(rng.randint(0, xs + 1, size=2))
        y0, y1 = sorted(rng.randint(0, ys + 1, size=2))
        m[x0:x1, y0:y1] = -1
    else:                                                  # scattered buildings, some of them on the border rows / columns
        m[rng.rand(xs, ys) < rng.uniform(0.05, 0.3)] = -1
        if kind == 3:
            if rng.rand() < 0.5:
                m[0, rng.rand(ys) < 0.5] = -1
            if rng.rand() < 0.5:
                m[:, 0][rng.rand(xs) < 0.5] = -1
            if rng.rand() < 0.3:
                m[-1, :] = -1
    while (m == 0).sum() * 3 < xs * ys:                    # keep room for the agents
        bx, by = np.nonzero(m)
        k = rng.randint(len(bx))
        m[bx[k], by[k]] = 0
    return m


def draw_case(rng):
    xs, ys = int(rng.randint(4, 29)), int(rng.randint(4, 29))
    if rng.rand() < 0.3:
        ys = xs
    n_maps = 1 if rng.rand() < 0.7 else int(rng.randint(2, 4))
    maps = [random_map(rng, xs, ys) for _ in range(n_maps)]
    big = rng.rand() < 0.15                                # more agents than one wavefront has lanes
    P = int(rng.randint(30, 50)) if big else int(rng.randint(1, 17))
    E = int(rng.randint(30, 60)) if big else int(rng.randint(1, 33))
    flatten = bool(rng.rand() < 0.65)
    cfg = dict(n_pursuers=P, n_evaders=E, obs_range=int(rng.choice([3, 4, 5, 6, 7, 7, 9, 11, 13])),
               n_catch=int(rng.randint(1, 5)), surround=bool(rng.rand() < 0.6), flatten=flatten,
               reward_mech=str(rng.choice(["local", "global"])),
               catchr=float(rng.choice([0.01, 0.1, 0.5, 1.0])), term_pursuit=float(rng.choice([5.0, 1.0, 0.0])),
               urgency_reward=float(rng.choice([0.0, -0.1, -0.05, 0.25])))
    if flatten and rng.rand() < 0.3:
        cfg["include_id"] = False
    if n_maps > 1:
        cfg["sample_maps"] = True
    if rng.rand() < 0.2 and E >= 3:
        cfg["random_opponents"] = True
        cfg["max_opponents"] = int(rng.randint(2, E + 2))   # n_evaders >= max_opponents - 1 (the reference indexes evaders_gone by slot, :138)
    run = dict(episodes=int(rng.randint(2, 5)), steps_per_episode=int(rng.randint(8, 25)), seed=int(rng.randint(1 << 30)),
               chase=float(rng.uniform(0.4, 0.95)), in_building=bool(rng.rand() < 0.12 and n_maps == 1))
    return maps, cfg, run


def main():
    R = ref_loader.load()
    rng = np.random.RandomState(MASTER_SEED)
    for i in range(N_CASES):
        maps, cfg, run = draw_case(rng)      # always drawn, so that case i is the same whichever subset is regenerated
        run_scenario(R, "fuzz_%02d" % i, maps, cfg, **run)


if __name__ == "__main__":
    main()
