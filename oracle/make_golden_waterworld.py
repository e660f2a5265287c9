#!/usr/bin/env python
"""Generate tests/golden/waterworld_*.npz by running the UNMODIFIED reference MAWaterWorld
(/root/reference/madrl_environments/pursuit/waterworld.py) under the shims in oracle/shims.

TEST INFRASTRUCTURE ONLY (build container; outputs are committed).

Protocol ("teacher forcing"): for every step the file records the
reference's float64 state BEFORE the step, the action, which evaders / poisons were caught,
the outcome of every respawn (accepted position + the two velocity uniforms, obtained by
instrumenting `env._respawn`, `env._caught` and `env.np_random.rand` at run time -- the
reference source is not modified), and the state, observations, rewards, done and info AFTER
the step.  A checker loads the pre-state, applies the same action and respawn outcomes and must
reproduce the post-state and outputs.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_loader  # noqa: E402

OUT = os.path.join(os.environ.get("MADRL_GOLDEN_OUT") or os.path.join(os.path.dirname(HERE), "tests", "golden"))


class LoggingRandom(object):
    def __init__(self, rs):
        self.rs = rs
        self.log = []

    def rand(self, *a):
        v = self.rs.rand(*a)
        self.log.append(("rand", np.array(v, dtype=np.float64).reshape(-1).copy()))
        return v

    def __getattr__(self, k):
        return getattr(self.rs, k)


def state_of(env):
    P = np.array([p.position for p in env._pursuers]); PV = np.array([p.velocity for p in env._pursuers])
    E = np.array([p.position for p in env._evaders]); EV = np.array([p.velocity for p in env._evaders])
    O = np.array([p.position for p in env._poisons]); OV = np.array([p.velocity for p in env._poisons])
    pos = np.concatenate([P, E, O]).astype(np.float64)
    vel = np.concatenate([PV, EV, OV]).astype(np.float64)
    return pos.copy(), vel.copy(), np.array(env.obstaclesx_No_2[0], dtype=np.float64).copy(), int(env._timesteps)


def run_scenario(R, name, ctor_args, ctor_kw, episodes, steps, seed, action_kind="uniform", cluster=False):
    MAWaterWorld = R["MAWaterWorld"]
    env = MAWaterWorld(*ctor_args, **ctor_kw)
    env.seed(seed)
    lr = LoggingRandom(env.np_random)
    env.np_random = lr
    events = []
    orig_caught, orig_respawn = env._caught, env._respawn

    def caught_logged(m, n_coop):
        out = orig_caught(m, n_coop)
        events.append(("caught", np.array(out[0]).copy(), np.array(out[1]).copy()))
        return out

    def respawn_logged(objx, radius):
        lr.log.append(("respawn_begin",))
        out = orig_respawn(objx, radius)
        lr.log.append(("respawn_end", np.array(out, dtype=np.float64).copy()))
        return out

    env._caught = caught_logged
    env._respawn = respawn_logged
    Np, Ne, Npo = env.n_pursuers, env.n_evaders, env.n_poison
    NP = Np + Ne + Npo
    rng = np.random.RandomState(seed + 1000)
    rec = dict(pre_pos=[], pre_vel=[], pre_t=[], obst=[], act=[], ev_caught=[], po_caught=[], resp=[],
               post_pos=[], post_vel=[], post_t=[], obs=[], rew=[], done=[], evc=[], poc=[], is_reset_step=[])

    def parse_respawns():
        """[(accepted_pos(2), vel_uniforms(2))] in call order from the rand/respawn log of one step."""
        out = []
        i = 0
        L = lr.log
        while i < len(L):
            if L[i][0] == "respawn_begin":
                j = i + 1
                while L[j][0] != "respawn_end":
                    j += 1
                acc = L[j][1]
                assert L[j + 1][0] == "rand" and L[j + 1][1].shape == (2,)
                out.append((acc, L[j + 1][1]))
                i = j + 2
            else:
                i += 1
        return out

    for ep in range(episodes):
        # reset(): replicate its particle initialisation, then record its trailing zero-action step
        # as a normal step record (W11).  We let the reference do reset() and capture the pre-state of
        # the inner step by intercepting step().
        inner = {}
        orig_step = env.step

        def step_spy(a, _orig=orig_step):
            inner["pre"] = state_of(env)
            del lr.log[:]
            del events[:]
            return _orig(a)

        env.step = step_spy
        obs = env.reset()
        env.step = orig_step
        if cluster:
            pass
        pre = inner["pre"]
        _record(rec, env, pre, np.zeros((Np, 2)), events, parse_respawns(), obs, None, None, None, True, Ne, Npo)
        # the reset's inner step returned only obs; rewards/done/info of that step are discarded by reset()
        for t in range(steps):
            if cluster and t % 7 == 0:
                # drag evaders / poisons next to pursuers so that catches happen often
                for k, ev in enumerate(env._evaders[:4]):
                    tgt = env._pursuers[k % Np].position
                    ev.set_position(np.clip(tgt + rng.uniform(-0.03, 0.03, 2), 0, 1))
                for k, po in enumerate(env._poisons[:2]):
                    tgt = env._pursuers[(k + 2) % Np].position
                    po.set_position(np.clip(tgt + rng.uniform(-0.02, 0.02, 2), 0, 1))
                if t % 14 == 0:
                    # and herd pursuers together for cooperative catches
                    c = env._pursuers[0].position
                    for pu in env._pursuers[1:3]:
                        pu.set_position(np.clip(c + rng.uniform(-0.02, 0.02, 2), 0, 1))
            if action_kind == "uniform":
                a = rng.uniform(-1, 1, size=(Np, 2))
            else:
                a = rng.randn(Np, 2) * 0.5
            pre = state_of(env)
            del lr.log[:]
            del events[:]
            obs, rew, done, info = env.step(a.reshape(-1) if t % 2 else a)
            _record(rec, env, pre, a, events, parse_respawns(), obs, rew, done, info, False, Ne, Npo)
    out = {k: np.asarray(v) for k, v in rec.items()}
    cfg = dict(n_pursuers=Np, n_evaders=Ne, n_coop=env.n_coop, n_poison=Npo, n_sensors=env.n_sensors,
               addid=int(env._addid), speed_features=int(env._speed_features),
               reward_global=int(env.reward_mech == "global"),
               obstacle_fixed=int(env.obstacle_loc is not None))
    for k, v in cfg.items():
        out["cfg_" + k] = np.int64(v)
    for k in ("radius", "obstacle_radius", "ev_speed", "poison_speed", "action_scale", "poison_reward",
              "food_reward", "encounter_reward", "control_penalty"):
        out["cfg_" + k] = np.float64(getattr(env, k))
    out["cfg_sensor_range"] = np.float64(env.sensor_range[0])
    out["cfg_obstacle_loc"] = np.asarray(env.obstacle_loc if env.obstacle_loc is not None else [np.nan, np.nan], dtype=np.float64)
    out["sensors"] = np.asarray(env._pursuers[0].sensors, dtype=np.float64)
    path = os.path.join(OUT, "waterworld_%s.npz" % name)
    np.savez_compressed(path, **out)
    print("%-26s steps=%4d evcatches=%3d pocatches=%3d respawns=%3d  %6.1f KB" % (
        name, len(out["pre_t"]), int(np.nansum(out["evc"])), int(np.nansum(out["poc"])),
        int((out["resp"][..., 0] > -1).sum()), os.path.getsize(path) / 1024.0))


def _record(rec, env, pre, a, events, respawns, obs, rew, done, info, is_reset, Ne, Npo):
    Np = env.n_pursuers
    NP = Np + Ne + Npo
    assert len(events) == 3 and events[0][0] == "caught"
    ev_caught, po_caught = events[0][1], events[1][1]
    assert len(respawns) == len(ev_caught) + len(po_caught)
    resp = -np.ones((NP, 4))  # per particle: accepted x, y, velocity uniforms u0, u1 (-1: not respawned)
    k = 0
    for e in ev_caught:
        resp[Np + e, :2] = respawns[k][0]; resp[Np + e, 2:] = respawns[k][1]; k += 1
    for p in po_caught:
        resp[Np + Ne + p, :2] = respawns[k][0]; resp[Np + Ne + p, 2:] = respawns[k][1]; k += 1
    evm = np.zeros(Ne, np.uint8); evm[ev_caught] = 1
    pom = np.zeros(Npo, np.uint8); pom[po_caught] = 1
    post = state_of(env)
    rec["pre_pos"].append(pre[0]); rec["pre_vel"].append(pre[1]); rec["pre_t"].append(pre[3]); rec["obst"].append(pre[2])
    rec["act"].append(np.asarray(a, dtype=np.float64).reshape(Np, 2))
    rec["ev_caught"].append(evm); rec["po_caught"].append(pom); rec["resp"].append(resp)
    rec["post_pos"].append(post[0]); rec["post_vel"].append(post[1]); rec["post_t"].append(post[3])
    rec["obs"].append(np.stack([np.asarray(o, dtype=np.float64) for o in obs]))
    rec["rew"].append(np.full(Np, np.nan) if rew is None else np.asarray(rew, dtype=np.float64))
    rec["done"].append(-1 if done is None else int(bool(done)))
    rec["evc"].append(np.nan if info is None else info["evcatches"])
    rec["poc"].append(np.nan if info is None else info["pocatches"])
    rec["is_reset_step"].append(int(is_reset))


def main():
    R = ref_loader.load()
    os.makedirs(OUT, exist_ok=True)
    # BASELINE C3 configuration: MAWaterWorld(5, 10) defaults (waterworld.py:77-81, :483)
    run_scenario(R, "c3_default", (5, 10), {}, episodes=3, steps=120, seed=1)
    # the same with frequent (cooperative) catches / poison hits / respawns
    run_scenario(R, "c3_catches", (5, 10), {}, episodes=3, steps=120, seed=2, cluster=True)
    # runner configuration: random obstacle (run_waterworld.py:41), global reward, gaussian actions
    run_scenario(R, "global_randobst", (5, 10), dict(obstacle_loc=None, reward_mech="global"), episodes=3,
                 steps=80, seed=3, action_kind="gauss", cluster=True)
    # rllab_gru_test.py:?? shape MAWaterWorld(3, 10, 2, 5), fewer sensors, no speed features, no id
    run_scenario(R, "small_nospeed", (3, 10, 2, 5), dict(n_sensors=12, speed_features=False, addid=False,
                                                         sensor_range=0.3), episodes=3, steps=80, seed=4, cluster=True)
    # n_coop = 1, large action scale (wall clipping), fast evaders leaving the arena (W6)
    run_scenario(R, "coop1_fast", (4, 6, 1, 7), dict(ev_speed=0.05, poison_speed=0.03, action_scale=0.05,
                                                     n_sensors=20), episodes=2, steps=150, seed=5, cluster=True)


if __name__ == "__main__":
    main()
