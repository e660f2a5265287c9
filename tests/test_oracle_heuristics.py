"""CPU tests (-m "not gpu"): the NumPy policy oracle against golden outputs of the unmodified reference policies."""
import os

import numpy as np

from oracle import heuristics_oracle as ho

G = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "heuristics.npz"))


def test_pursuit_policy_oracle():
    for R in (7, 5, 11):
        a = ho.pursuit_actions(G["pursuit_R%d_obs" % R])
        assert np.array_equal(a, G["pursuit_R%d_act" % R]), R
        assert (a == -1).sum() > 50 and len(np.unique(a)) == 6


def test_waterworld_policy_oracle():
    a = ho.waterworld_actions(G["waterworld_obs"])
    assert np.abs(a - G["waterworld_act"]).max() < 1e-12
    assert np.abs(a[-50:]).max() == 0.0


def test_multiwalker_policy_oracle():
    a = ho.multiwalker_actions(G["multiwalker_obs"])
    assert np.abs(a - G["multiwalker_act"]).max() < 1e-12
