"""Shared helpers for the parity tests (golden replay, oracle <-> HIP state conversion)."""
import glob
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pursuit_golden_files():
    return sorted(glob.glob(os.path.join(GOLDEN, "pursuit_*.npz")))


def golden_id(path):
    return os.path.basename(path)[:-4]


def evadercontrol_golden_files():
    return sorted(glob.glob(os.path.join(GOLDEN, "evadercontrol_*.npz")))


def replay_evadercontrol(g, reset_fn, step_fn, state_fn):
    """Drive one env through an evadercontrol_*.npz record of the unmodified reference (train_pursuit=False) and compare
    everything it returned.  reset_fn(pos[1, P+E, 2]) -> obs rows; step_fn(agent_actions[1, P], pursuer_actions[1, P]) ->
    (obs rows, rew, done, removed); state_fn() -> dict(pos_p, pos_e, gone), all for one env as numpy arrays."""
    P = int(g["cfg_n_pursuers"])
    n_none = 0
    for t in range(len(g["op"])):
        if g["op"][t] == 0:
            obs = reset_fn(np.concatenate([g["init_p"][t], g["init_e"][t]])[None])
        else:
            obs, rew, done, rem = step_fn(g["act_a"][t][None], g["act_o"][t][None])
            np.testing.assert_array_equal(np.asarray(rew, np.float64).reshape(-1), g["rew_f64"][t].astype(np.float32 if np.asarray(rew).dtype == np.float32 else np.float64),
                                          err_msg="rewards, op %d" % t)
            assert int(done) == int(g["done"][t]) and int(rem) == int(g["removed"][t]), "done / removed, op %d" % t
        st = state_fn()
        np.testing.assert_array_equal(st["pos_p"], g["pos_p"][t], err_msg="pursuers, op %d" % t)
        np.testing.assert_array_equal(st["pos_e"], g["pos_e"][t], err_msg="evaders, op %d" % t)
        np.testing.assert_array_equal(st["gone"], g["gone_e"][t])
        none = g["obs_none"][t]
        assert np.array_equal(none, g["gone_e"][t][:P]), "None entries are the gone evader slots below n_pursuers (:418-428)"
        n_rows = P - int(none.sum())
        got = np.asarray(obs).reshape(P, -1)
        np.testing.assert_array_equal(got[:n_rows], g["obs_f32"][t][:n_rows], err_msg="observation rows, op %d" % t)
        n_none += int(none.sum())
    assert n_none > 0 and int(g["removed"].sum()) > 0
