"""CPU tests (-m "not gpu"): the NumPy wrapper oracle against golden outputs of the unmodified
reference wrappers (oracle/make_golden_wrappers.py)."""
import os

import numpy as np

from oracle import wrappers_oracle as wo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden", "wrappers_replay.npz")


def test_standardized_env_oracle_matches_reference():
    g = np.load(G)
    T, P, D = g["obs"].shape
    o = wo.StdOracle((P, D), (P,), scale_reward=float(g["std_cfg_scale_reward"]), enable_obsnorm=True, enable_rewnorm=True,
                     obs_alpha=float(g["std_cfg_obs_alpha"]), rew_alpha=float(g["std_cfg_rew_alpha"]), eps=float(g["std_cfg_eps"]))
    for t in range(T):
        so = o.obs(g["obs"][t].astype(np.float64))
        assert np.abs(so - g["std_obs"][t]).max() < 1e-12
        if g["op"][t] == 1:
            assert np.abs(o.rew(g["rew"][t]) - g["std_rew"][t]).max() < 1e-12


def test_observation_buffer_oracle_matches_reference():
    g = np.load(G)
    T, P, D = g["obs"].shape
    b = wo.BufOracle((P, D), int(g["buf_k"]))
    for t in range(T):
        out = b.reset(g["obs"][t]) if g["op"][t] == 0 else b.step(g["obs"][t])
        assert np.array_equal(out.astype(np.float32), g["buf_obs"][t])


def test_diagnostics_oracle_matches_reference():
    g = np.load(G)
    T, P, D = g["obs"].shape
    d = wo.DiagOracle(1, P, discount=float(g["diag_discount"]), max_traj_len=int(g["diag_max_traj_len"]))
    k = 0
    for t in range(T):
        if g["op"][t] == 0:
            d.reset()
            continue
        out = d.step(g["rew"][t][None], np.array([g["done"][t]]))
        if out["finished"][0]:
            assert t == g["diag_at"][k]
            assert np.abs(out["reward"][0] - g["diag_reward"][k]).max() < 1e-12
            assert abs(out["disc"][0] - g["diag_disc"][k]) < 1e-12 and out["length"][0] == g["diag_len"][k]
            assert abs(out["reward"][0].mean() - g["diag_avg"][k]) < 1e-12
            k += 1
    assert k == len(g["diag_at"]) and k >= 5
