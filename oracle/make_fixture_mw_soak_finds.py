#!/usr/bin/env python
"""tests/fixtures/mw_soak_finds.npz: the two env-steps on which scripts/mw_soak.py (other seeds than the committed soak) caught the product's
MultiWalker source disagreeing with the independent restatement -- for each: the product's raw world record of that env
just before the step, the step's actions, and what the INDEPENDENT restatement (oracle/multiwalker_ref.c) has after it.  TEST INFRASTRUCTURE
ONLY; needs no reference tree (both sides are in this repository): it re-runs the two soaks up to the step (about a minute).
    1. six walkers, seed 1106, env 41, step 1998: b2ContactManager::AddPair wakes a hull the step's islands had just put to sleep
    2. ten walkers, seed 5110, env 20, step 3133: b2Contact::Update re-enables a contact an earlier event of the continuous pass disabled
tests/test_multiwalker_cpu.py replays them on the CPU build of the product source."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import multiwalker as mwo, multiwalker_ref as mwr  # noqa: E402


def capture(W, seed, env, step):
    N = 64
    ref = mwr.MultiWalkerRef(n_walkers=W, n_envs=N, seed=seed, position_noise=0, angle_noise=0, poly=True)
    core = mwo.MultiWalkerOracle(n_walkers=W, n_envs=N, seed=seed, position_noise=0.0, angle_noise=0.0)
    ref.reset(); core.reset()
    rng = np.random.RandomState(seed)
    for t in range(step + 1):
        a = rng.uniform(-1, 1, (N, W, 4)).astype(np.float32)
        if (t // 40) % 5 == 4:
            a[:] = 0
        if t == step:
            assert np.array_equal(ref.bodies(), core.bodies()[0]) and np.array_equal(ref.aux(), core.aux()), "the two sides must agree up to the step"
            before = core.worlds()[env].copy()
        ro, rr, rd = ref.step(a); core.step(a)
        if t < step and rd.any():
            ref.reset(mask=rd); core.reset(mask=rd)
    return dict(world=before, actions=a[env].copy(), bodies=ref.bodies()[env].copy(), aux=ref.aux()[env].copy(), joints=ref.joints()[env].copy(),
                flags=ref.flags()[env].copy(), n_walkers=np.int64(W), seed=np.int64(seed), env=np.int64(env), step=np.int64(step))


def main():
    out = {}
    for name, args in (("addpair_wakes", (6, 1106, 41, 1998)), ("update_reenables", (10, 5110, 20, 3133))):
        for k, v in capture(*args).items():
            out["%s_%s" % (name, k)] = v
    path = os.path.join(os.path.dirname(HERE), "tests", "fixtures", "mw_soak_finds.npz")
    np.savez_compressed(path, **out)
    print("%s: %.1f KB" % (path, os.path.getsize(path) / 1024.0))


if __name__ == "__main__":
    main()
