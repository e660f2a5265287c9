#!/usr/bin/env python
"""Generate tests/golden/hostage_*.npz by running the UNMODIFIED reference ContinuousHostageWorld
(/root/reference/madrl_environments/hostage.py) under the shims in oracle/shims.  TEST INFRASTRUCTURE ONLY.

Teacher-forcing protocol (as for Waterworld): every record holds the reference's float64 state BEFORE a step, the action,
the four uniforms of every criminal respawn of that step (logged from `env.np_random.rand` at run time; the source is not
modified), and the state, observations, rewards, done and info AFTER it.  reset() is recorded through its trailing
zero-action step (hostage.py:177); its sampling statements are checked separately on the Philox side."""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_loader  # noqa: E402

OUT = os.path.join(os.environ.get("MADRL_GOLDEN_OUT") or os.path.join(os.path.dirname(HERE), "tests", "golden"))


class LoggingRandom(object):
    def __init__(self, rs):
        self.rs, self.log = rs, []

    def rand(self, *a):
        v = self.rs.rand(*a)
        self.log.append(np.array(v, dtype=np.float64).reshape(-1).copy())
        return v

    def __getattr__(self, k):
        return getattr(self.rs, k)


def state_of(env):
    ag = list(env._rescuers) + list(env._hostages) + list(env._criminals)
    pos = np.array([a.position for a in ag], dtype=np.float64)
    vel = np.array([a.velocity for a in ag], dtype=np.float64)
    return dict(pos=pos.copy(), vel=vel.copy(), key=np.array(env.key_loc, np.float64).reshape(2).copy(),
                bomb=np.array(env.bomb_loc, np.float64).reshape(2).copy(), saved=np.array(env.curr_host_saved_mask, np.uint8).copy(),
                gate=int(env._gate_open), bombed=int(env._bombed), t=int(env._timesteps))


def run(H, name, args, kw, episodes, steps, seed, herd=True):
    env = H.ContinuousHostageWorld(*args, **kw)
    env.seed(seed)
    lr = LoggingRandom(env.np_random)
    env.np_random = lr
    Nr, Nh, Nc = env.n_good, env.n_hostages, env.n_bad
    rng = np.random.RandomState(seed + 100)
    rec = {k: [] for k in ("pre_pos", "pre_vel", "pre_saved", "pre_gate", "pre_bombed", "pre_t", "key", "bomb", "act", "resp", "post_pos",
                           "post_vel", "post_saved", "post_gate", "post_bombed", "post_t", "obs", "rew", "done", "info", "is_reset_step")}

    def record(pre, a, obs, rew, done, info, is_reset):
        post = state_of(env)
        # respawns of this step: criminals caught by >= 1 rescuer (pre-respawn collision), ascending index, rand(2) + rand(2) each
        log = [v for v in lr.log]
        resp = -np.ones((Nc, 4))
        k = 0
        # identify respawned criminals exactly: recompute the collision test of the reference on the post-move rescuer positions
        resc = post["pos"][:Nr]
        for j in range(Nc):
            cr = pre["pos"][Nr + Nh + j]
            d = np.sqrt(((resc - cr[None]) ** 2).sum(axis=1))
            if (d <= env.radius + env.radius).any():
                resp[j, :2] = log[k]; resp[j, 2:] = log[k + 1]; k += 2
        assert k == len(log), (k, len(log))
        rec["pre_pos"].append(pre["pos"]); rec["pre_vel"].append(pre["vel"]); rec["pre_saved"].append(pre["saved"])
        rec["pre_gate"].append(pre["gate"]); rec["pre_bombed"].append(pre["bombed"]); rec["pre_t"].append(pre["t"])
        rec["key"].append(pre["key"]); rec["bomb"].append(pre["bomb"]); rec["act"].append(np.asarray(a, np.float64).reshape(Nr, 2))
        rec["resp"].append(resp)
        rec["post_pos"].append(post["pos"]); rec["post_vel"].append(post["vel"]); rec["post_saved"].append(post["saved"])
        rec["post_gate"].append(post["gate"]); rec["post_bombed"].append(post["bombed"]); rec["post_t"].append(post["t"])
        rec["obs"].append(np.stack([np.asarray(o, np.float64) for o in obs]))
        rec["rew"].append(np.full(Nr, np.nan) if rew is None else np.asarray(rew, np.float64))
        rec["done"].append(-1 if done is None else int(bool(done)))
        rec["info"].append([-1, -1] if info is None else [info["ho_saved"], info["cr_encs"]])
        rec["is_reset_step"].append(int(is_reset))

    for ep in range(episodes):
        inner = {}
        orig_step = env.step

        def spy(a, _orig=orig_step):
            inner["pre"] = state_of(env)
            del lr.log[:]
            return _orig(a)

        env.step = spy
        obs = env.reset()
        env.step = orig_step
        record(inner["pre"], np.zeros((Nr, 2)), obs, None, None, None, True)
        for t in range(steps):
            if herd:
                # drive the scenario through its phases: to the key, then through the gate to the hostages, criminals in the way
                if t == 5:
                    env._rescuers[0].set_position(np.clip(np.squeeze(env.key_loc) - 0.004, 0, 1))
                if t > 8 and t % 9 == 0:
                    tgt = env._hostages[(t // 9) % Nh].position
                    for r_ in env._rescuers[:max(env.n_coop_save, 1)]:
                        r_.set_position(np.clip(tgt + rng.uniform(-0.01, 0.01, 2), 0, 1))
                if t % 7 == 3:
                    env._criminals[t % Nc].set_position(np.clip(env._rescuers[t % Nr].position + rng.uniform(-0.01, 0.01, 2), 0, 1))
                if t % 31 == 30:
                    env._criminals[0].set_position(np.array([0.999, 0.999])); env._criminals[0].set_velocity(np.array([0.004, 0.003]))
                if ep == episodes - 1 and t == steps - 20:
                    env._rescuers[-1].set_position(np.squeeze(env.bomb_loc) + 0.01)
            a = rng.uniform(-1, 1, size=(Nr, 2)) if t % 3 else rng.randn(Nr, 2) * 2
            pre = state_of(env)
            del lr.log[:]
            obs, rew, done, info = env.step(a.reshape(-1) if t % 2 else a)
            record(pre, a, obs, rew, done, info, False)
            if done:
                break
    out = {k: np.asarray(v) for k, v in rec.items()}
    for k in ("n_good", "n_hostages", "n_bad", "n_coop_save", "n_coop_avoid", "n_sensors"):
        out["cfg_" + k] = np.int64(getattr(env, k))
    out["cfg_addid"] = np.int64(env._addid); out["cfg_reward_global"] = np.int64(env.reward_mech == "global")
    for k in ("radius", "bad_speed", "action_scale", "save_reward", "hit_reward", "encounter_reward", "not_saved_reward", "bomb_reward",
              "bomb_radius", "key_radius", "control_penalty"):
        out["cfg_" + k] = np.float64(getattr(env, k))
    out["cfg_sensor_range"] = np.float64(env.sensor_range[0])
    out["sensors"] = np.asarray(env._rescuers[0].sensors, np.float64)
    path = os.path.join(OUT, "hostage_%s.npz" % name)
    np.savez_compressed(path, **out)
    print("%-18s steps=%4d respawns=%3d saved=%d gate=%d bombed=%d done=%d  %5.1f KB" % (
        name, len(out["pre_t"]), int((out["resp"][..., 0] >= 0).sum()), int(out["post_saved"].max(axis=0).sum()), int(out["post_gate"].max()),
        int(out["post_bombed"].max()), int((out["done"] == 1).sum()), os.path.getsize(path) / 1024.0))


def main():
    ref_loader.load()
    H = importlib.import_module("madrl_environments.hostage")
    # the module's own example configuration (hostage.py:483) and the runner's default reward mechanism (global)
    run(H, "default_global", (3, 10, 5, 2, 2), {}, episodes=3, steps=150, seed=1)
    run(H, "local", (3, 10, 5, 2, 2), dict(reward_mech="local"), episodes=3, steps=150, seed=2)
    run(H, "small_noid", (2, 3, 2, 1, 1), dict(reward_mech="local", addid=False, n_sensors=12, sensor_range=0.3, bad_speed=0.03), episodes=3,
        steps=120, seed=3)
    run(H, "free", (4, 6, 4, 2, 1), dict(action_scale=0.03), episodes=2, steps=150, seed=4, herd=False)


if __name__ == "__main__":
    main()
