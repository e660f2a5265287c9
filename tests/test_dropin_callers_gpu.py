"""GPU test (-m gpu): the N == 1 drop-in PursuitEvade driven by the reference's own callers.

tests/golden/callers_pursuit.npz was recorded by oracle/make_golden_callers.py running UNMODIFIED reference code --
AbstractMAEnv.animate (madrl_environments/__init__.py:72-107), DiagnosticsWrapper(StandardizedEnv(env))
(:204-311, :314-369) and the rollout loop of heuristics/pursuit.py:64-85 -- over the reference env.  Here the same calls
run over madrl_amd.pursuit.PursuitEvade / madrl_amd.wrappers and every returned VALUE is compared, not just its type."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden", "callers_pursuit.npz")
KW = dict(n_evaders=30, n_pursuers=8, obs_range=7, n_catch=2, surround=True, flatten=True, reward_mech="local")


class QueueController(object):
    """the evader_controller of the recording: one flat list, one act() per remaining evader (pursuit_evade.py:238-241)"""

    def __init__(self, actions):
        self.actions, self.used = np.asarray(actions), 0

    def act(self, state):
        assert state.shape == (4, 16, 16) and state.dtype == np.float32
        a = int(self.actions[self.used])
        self.used += 1
        return a


def act_fn(o):
    return int(np.floor(np.sum(np.asarray(o, dtype=np.float64)) * 7.0)) % 5


def _env(g, prefix, **kw):
    from madrl_amd.pursuit import PursuitEvade
    env = PursuitEvade([g["map"].astype(np.int32)], evader_controller=QueueController(g[prefix + "_eacts"]), **kw)
    env.script_reset_positions(list(g[prefix + "_pos"]))
    return env


def test_animate_loop_matches_the_reference():
    g = np.load(G)
    env = _env(g, "a", **KW)
    seen = []

    def logging_act_fn(o):
        assert isinstance(o, np.ndarray) and o.dtype == np.float64
        seen.append(o.copy())
        return act_fn(o)

    rew, traj_info = env.animate(logging_act_fn, int(g["a_nsteps"]))
    got = np.stack(seen).reshape(-1, 8, seen[0].shape[0])
    assert got.shape == g["a_obs"].shape, "the loop stopped at a different step"
    assert np.array_equal(got.astype(np.float32), g["a_obs"]), "observations handed to the policy functions"
    assert np.array_equal(traj_info["removed"], g["a_removed"])
    assert np.abs(np.asarray(rew) - g["a_rew"]).max() < 1e-6 * max(1.0, np.abs(g["a_rew"]).max())  # float32 reward outputs summed in float64
    assert env._evader_controller.used == int(np.sum([30 - np.sum(g["a_removed"][:t]) for t in range(len(g["a_removed"]))]))


def test_standardized_diagnostics_stack_matches_the_reference():
    from madrl_amd.wrappers import StandardizedEnv, DiagnosticsWrapper
    g = np.load(G)
    env = _env(g, "b", **KW)
    cfg = {k[6:]: float(g[k]) for k in g.files if k.startswith("b_cfg_")}
    cfg["enable_obsnorm"], cfg["enable_rewnorm"] = bool(cfg["enable_obsnorm"]), bool(cfg["enable_rewnorm"])
    w = DiagnosticsWrapper(StandardizedEnv(env, **cfg), discount=float(g["b_discount"]), max_traj_len=int(g["b_max_traj_len"]))
    t_act = 0
    closed = 0
    for i, op in enumerate(g["b_op"]):
        if op == 0:
            obs = w.reset()
            assert isinstance(obs, list) and len(obs) == 8
        else:
            obs, rew, done, log = w.step(g["b_pacts"][t_act])
            t_act += 1
            assert isinstance(rew, list) and isinstance(done, bool) and isinstance(log, dict)
            assert done == bool(g["b_done"][i])
            assert np.abs(np.asarray(rew) - g["b_rew"][i]).max() < 1e-5 * max(1.0, np.abs(g["b_rew"][i]).max()), "op %d rewards" % i
            if np.isnan(g["b_log"][i][0]):
                assert "global/episode_length" not in log
            else:
                closed += 1
                want = g["b_log"][i]
                assert log["global/episode_length"] == int(want[2])
                assert abs(log["global/episode_avg_reward"] - want[0]) < 1e-4 * max(1.0, abs(want[0]))
                assert abs(log["global/episode_disc_return"] - want[1]) < 1e-4 * max(1.0, abs(want[1]))
                assert abs(log["global/episode_reward_agent3"] - want[3]) < 1e-4 * max(1.0, abs(want[3]))
        assert np.abs(np.stack(obs) - g["b_obs"][i]).max() < 1e-5, "op %d standardised observations" % i
    assert closed >= 2


def test_heuristic_rollout_loop_matches_the_reference():
    """heuristics/pursuit.py:64-85: (R, R, 4) observations, n_catch 4 without surround; the recorded actions of the
    reference policy are replayed (its fallback draws come from an unseeded generator)."""
    g = np.load(G)
    env = _env(g, "c", n_evaders=30, n_pursuers=8, obs_range=7, n_catch=4, surround=False, flatten=False)
    obs = env.reset()
    assert np.array_equal(np.stack(obs).astype(np.float32), g["c_obs"][0]) and obs[0].shape == (7, 7, 4)
    total = 0.0
    for t in range(len(g["c_act"])):
        obs, r, done, info = env.step(list(g["c_act"][t]))
        total += np.mean(r)
        assert np.array_equal(np.stack(obs).astype(np.float32), g["c_obs"][t + 1]), "step %d observations" % t
        assert np.array_equal(np.asarray(r, dtype=np.float32), g["c_rew"][t].astype(np.float32)), "step %d rewards" % t
        assert done == bool(g["c_done"][t]) and info == {"removed": int(g["c_removed"][t])}
        if done:
            break
    assert abs(total - float(g["c_total"])) < 1e-5
