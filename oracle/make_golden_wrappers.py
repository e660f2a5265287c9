#!/usr/bin/env python
"""Golden vectors for the env wrappers: the UNMODIFIED reference
StandardizedEnv / ObservationBuffer / DiagnosticsWrapper (madrl_environments/__init__.py:143-389)
wrapped around a replay env that plays back a recorded PursuitEvade episode sequence
(tests/golden/pursuit_c1_surround_local.npz).  TEST INFRASTRUCTURE ONLY."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_loader  # noqa: E402

OUT = os.path.join(os.environ.get("MADRL_GOLDEN_OUT") or os.path.join(os.path.dirname(HERE), "tests", "golden"))


def main():
    R = ref_loader.load()
    M = R["madrl_environments"]
    from gym import spaces
    g = np.load(os.path.join(OUT, "pursuit_c1_surround_local.npz"))
    ops, obs, rew, done = g["op"], g["obs_f32"].astype(np.float64), g["rew_f64"], g["done"]
    T, P, D = obs.shape

    class ReplayAgent(M.Agent):
        @property
        def observation_space(self):
            return spaces.Box(low=-np.inf, high=np.inf, shape=(D,))

        @property
        def action_space(self):
            return spaces.Discrete(5)

    class ReplayEnv(M.AbstractMAEnv):
        def __init__(self):
            self.t = -1
            self._agents = [ReplayAgent() for _ in range(P)]

        @property
        def agents(self):
            return self._agents

        @property
        def reward_mech(self):
            return "local"

        def reset(self):
            self.t += 1
            assert ops[self.t] == 0
            return [obs[self.t, i].copy() for i in range(P)]

        def step(self, a):
            self.t += 1
            assert ops[self.t] == 1
            return [obs[self.t, i].copy() for i in range(P)], rew[self.t].copy(), bool(done[self.t]), {"removed": 0}

    out = dict(op=ops, obs=obs.astype(np.float32), rew=rew, done=done)
    # --- StandardizedEnv (:204-311)
    cfg = dict(scale_reward=0.5, enable_obsnorm=True, enable_rewnorm=True, obs_alpha=0.01, rew_alpha=0.02, eps=1e-8)
    env = M.StandardizedEnv(ReplayEnv(), **cfg)
    so, sr = [], []
    for t in range(T):
        if ops[t] == 0:
            o = env.reset(); r = [np.nan] * P
        else:
            o, r, d, info = env.step(None)
        so.append(np.stack(o)); sr.append(np.asarray(r, dtype=np.float64))
    out["std_obs"] = np.asarray(so); out["std_rew"] = np.asarray(sr)
    for k, v in cfg.items():
        out["std_cfg_" + k] = np.float64(v)
    # --- ObservationBuffer (:143-201); only reset/step are usable (the `agents` property has a typo)
    K = 4
    env = M.ObservationBuffer.__new__(M.ObservationBuffer)
    env._unwrapped = ReplayEnv(); env._buffer_size = K
    env._buffer = [np.zeros((D, K)) for _ in range(P)]
    bo = []
    for t in range(T):
        o = env.reset() if ops[t] == 0 else env.step(None)[0]
        bo.append(np.stack(o))
    out["buf_obs"] = np.asarray(bo, dtype=np.float32); out["buf_k"] = np.int64(K)
    # --- DiagnosticsWrapper (:314-389)
    disc, mtl = 0.97, 25
    env = M.DiagnosticsWrapper(ReplayEnv(), discount=disc, max_traj_len=mtl, log_interval=10**9)
    ep_reward, ep_avg, ep_disc, ep_len, ep_at = [], [], [], [], []
    for t in range(T):
        if ops[t] == 0:
            env.reset()
        else:
            o, r, d, log = env.step(None)
            if "global/episode_length" in log:
                ep_at.append(t)
                ep_reward.append([log["global/episode_reward_agent%d" % i] for i in range(P)])
                ep_avg.append(log["global/episode_avg_reward"]); ep_disc.append(log["global/episode_disc_return"])
                ep_len.append(log["global/episode_length"])
    out.update(diag_at=np.asarray(ep_at), diag_reward=np.asarray(ep_reward), diag_avg=np.asarray(ep_avg),
               diag_disc=np.asarray(ep_disc), diag_len=np.asarray(ep_len), diag_discount=np.float64(disc),
               diag_max_traj_len=np.int64(mtl))
    path = os.path.join(OUT, "wrappers_replay.npz")
    np.savez_compressed(path, **out)
    print("wrappers_replay: T=%d episodes logged=%d  %.1f KB" % (T, len(ep_at), os.path.getsize(path) / 1024.0))


if __name__ == "__main__":
    main()
