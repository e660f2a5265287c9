"""CPU test: the independent MultiWalker oracle (oracle/multiwalker_ref.c, libm sin / cos like Box2D) against recordings of the
UNMODIFIED reference MultiWalkerEnv running on real Box2D -- tests/golden/multiwalker_box2d_*.npz, written by
oracle/make_golden_multiwalker.py.

The build image has no Box2D, so those files do not exist in this repository and the test SKIPS: the MultiWalker oracles are PARITY
UNPINNED (their only anchor to Box2D itself is the published HelloWorld trace, tests/test_multiwalker_cpu.py).  With the files present
the oracle is teacher-forced on the recorded body states step by step and must reproduce the next recorded state, the observations and
the rewards within 1e-5, and the ContactDetector flags / done exactly."""
import glob
import os

import numpy as np
import pytest

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "multiwalker_box2d_*.npz")))
TOL = 1e-5


@pytest.mark.skipif(not GOLDEN, reason="PARITY UNPINNED: no Box2D in the build image, tests/golden/multiwalker_box2d_*.npz were never generated "
                                       "(oracle/make_golden_multiwalker.py writes them where pybox2d imports)")
@pytest.mark.parametrize("revision", [0, 1], ids=["b2CollidePolygons-2.3.0", "b2CollidePolygons-later-2.3.x"])
@pytest.mark.parametrize("path", GOLDEN or ["none"], ids=[os.path.basename(p) for p in GOLDEN] or ["none"])
def test_independent_oracle_reproduces_box2d_recordings(path, revision):
    """(Both revisions of b2CollidePolygons are tried: the one the recording's Box2D has passes, the other may fail on the first hull / package
    contact it resolves differently -- whichever passes is the `polygon_revision` / `box2d_polygon_revision` to run with.)"""
    from oracle import multiwalker_ref as mwr
    g = np.load(path)
    W = int(g["n_walkers"])
    for ep in range(int(g["n_episodes"])):
        k = lambda name: g["ep%d_%s" % (ep, name)]
        ref = mwr.MultiWalkerRef(n_walkers=W, n_envs=1, seed=0, position_noise=0, angle_noise=0, reward_mech="global" if int(g["reward_global"]) else "local",
                                 poly=False, polygon_revision=revision)
        obs = ref.reset(terrain=k("terrain_y")[None], push=k("push")[None])
        assert np.abs(ref.bodies()[0] - k("bodies")[0]).max() <= TOL, "episode %d: state after reset" % ep
        assert np.abs(obs[0] - k("obs")[0]).max() <= TOL
        for t in range(len(k("done"))):
            ref.set_bodies(k("bodies")[t][None])
            obs, rew, done = ref.step(k("actions")[t][None])
            assert np.abs(ref.bodies()[0] - k("bodies")[t + 1]).max() <= TOL, "episode %d step %d: body states" % (ep, t)
            assert np.array_equal(ref.flags()[0], k("flags")[t + 1]) and int(done[0]) == int(k("done")[t]), "episode %d step %d: flags / done" % (ep, t)
            assert np.abs(obs[0] - k("obs")[t + 1]).max() <= TOL * max(1.0, np.abs(k("obs")[t + 1]).max())
            assert np.abs(rew[0] - k("rew")[t]).max() <= TOL * max(1.0, np.abs(k("rew")[t]).max())
