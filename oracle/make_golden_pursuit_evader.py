#!/usr/bin/env python
"""Generate tests/golden/evadercontrol_*.npz: the UNMODIFIED reference PursuitEvade with train_pursuit=False
(pursuit_evade.py:105-112, :204-207, :215-224) -- the actions drive the evaders, the pursuers move by
`pursuer_controller`, the observations are the evaders' windows, the rewards stay the pursuers'.

TEST INFRASTRUCTURE ONLY (runs in the build container; outputs are committed).

Randomness is pinned as in make_golden_pursuit.py: initial positions by replaying a position list through
`agent_utils.feasible_position`, pursuer moves through a scripted `pursuer_controller=` (one act() per pursuer, :238-241).
The driver passes min(n_pursuers, remaining evaders) actions per step: `for i, a in enumerate(actions):
agent_layer.move_agent(i, a)` (:229-230) raises IndexError in the reference once fewer evaders than actions remain.

Record per op (0 = reset, 1 = step), one env object per file so that quirk Q2 carries across resets:
  obs_f32 [T, P, D]  row k = the k-th non-None entry of the returned list (the reference computes it in local_obs[k]);
  obs_none [T, P]    which list entries were None (evaders_gone[i] for i < P, collect_obs :418-428);
  act_a [T, P]       agent actions (entry k moves layer agent k; 4 where none was passed), act_o [T, P] pursuer actions;
  rew_f64 [T, P], done, removed, pos_p [T, P, 2], pos_e [T, E, 2] (slot order, -1 = gone), gone_e [T, E].
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_loader  # noqa: E402
from oracle.make_golden_pursuit import ScriptedController, free_cells  # noqa: E402

OUT = os.path.join(os.environ.get("MADRL_GOLDEN_OUT") or os.path.join(os.path.dirname(HERE), "tests", "golden"))


def run(R, name, maps, cfg, episodes, steps, seed):
    PursuitEvade = R["PursuitEvade"]
    from madrl_environments.pursuit.utils import agent_utils
    rng = np.random.RandomState(seed)
    maps = [np.asarray(m, dtype=np.int32) for m in maps]
    ctrl = ScriptedController()
    env = PursuitEvade(maps, train_pursuit=False, pursuer_controller=ctrl, **cfg)
    P, E = env.n_pursuers, env.n_evaders
    flatten = cfg.get("flatten", True)
    D = (3 * env.obs_range ** 2 + (1 if env.include_id else 0)) if flatten else 4 * env.obs_range ** 2
    pos_queue = []
    orig = agent_utils.feasible_position
    agent_utils.feasible_position = lambda map_matrix, constraints=None: pos_queue.pop(0)
    rec = {k: [] for k in ("op", "init_p", "init_e", "act_a", "act_o", "obs_f32", "obs_none", "rew_f64", "done", "removed",
                           "pos_p", "pos_e", "gone_e")}
    cast_err = 0.0

    def snapshot():
        pp = np.array([env.pursuer_layer.get_position(i).copy() for i in range(P)], dtype=np.int32)
        pe = -np.ones((E, 2), dtype=np.int32)
        k = 0
        for i in range(E):
            if not env.evaders_gone[i]:
                pe[i] = env.evader_layer.get_position(k)
                k += 1
        assert k == env.evader_layer.n_agents()
        return pp, pe, env.evaders_gone.astype(np.uint8).copy()

    def obs_rows(obslist):
        nonlocal cast_err
        assert len(obslist) == P
        rows, none = np.zeros((P, D), np.float32), np.zeros(P, np.uint8)
        k = 0
        for i, o in enumerate(obslist):
            if o is None:
                none[i] = 1
                continue
            o64 = np.array(o, dtype=np.float64, copy=True).reshape(-1)
            rows[k] = o64.astype(np.float32)
            cast_err = max(cast_err, float(np.abs(o64 - rows[k].astype(np.float64)).max()))
            k += 1
        return rows, none

    def push(op, ip, ie, aa, ao, obs, rew, done, removed):
        pp, pe, ge = snapshot()
        rows, none = obs_rows(obs)
        for k, v in (("op", op), ("init_p", ip), ("init_e", ie), ("act_a", aa), ("act_o", ao), ("obs_f32", rows), ("obs_none", none),
                     ("rew_f64", np.asarray(rew, dtype=np.float64)), ("done", int(bool(done))), ("removed", int(removed)),
                     ("pos_p", pp), ("pos_e", pe), ("gone_e", ge)):
            rec[k].append(v)
        return pp, pe

    try:
        for ep in range(episodes):
            np.random.seed(rng.randint(2 ** 31 - 1))
            cells = free_cells(maps[0])
            crowd = cells[:max(6, len(cells) // 8)] if ep % 2 == 1 else cells   # odd episodes: crowded corner, catches happen
            ip = crowd[rng.randint(len(crowd), size=P)]
            ie = crowd[rng.randint(len(crowd), size=E)]
            pos_queue[:] = [tuple(int(v) for v in p) for p in ip] + [tuple(int(v) for v in p) for p in ie]
            obs = env.reset()
            assert not pos_queue
            pp, pe = push(0, ip.astype(np.int32), ie.astype(np.int32), np.full(P, 4, np.int32), np.full(P, 4, np.int32), obs,
                          np.zeros(P), 0, 0)
            for t in range(steps):
                n_left = env.evader_layer.n_agents()
                if n_left == 0:
                    break
                n_act = min(P, n_left)
                aa = np.full(P, 4, np.int32)
                aa[:n_act] = np.where(rng.rand(n_act) < (0.6 if ep % 2 == 1 else 0.2), 4, rng.randint(5, size=n_act))
                ao = rng.randint(5, size=P).astype(np.int32)
                alive = [i for i in range(E) if not env.evaders_gone[i]]
                for j in range(P):   # pursuers chase the nearest remaining evader half of the time
                    if rng.rand() < 0.6:
                        d = pe[alive] - pp[j]
                        dx, dy = d[np.argmin(np.abs(d).sum(1))]
                        ao[j] = 4 if abs(dx) + abs(dy) <= 1 else ((1 if dx > 0 else 0) if abs(dx) >= abs(dy) else (2 if dy > 0 else 3))
                ctrl.load(ao)
                obs, rew, done, info = env.step([int(a) for a in aa[:n_act]])
                assert ctrl.used == P
                pp, pe = push(1, np.zeros((P, 2), np.int32), np.zeros((E, 2), np.int32), aa, ao, obs, rew, done, info["removed"])
    finally:
        agent_utils.feasible_position = orig
    out = {k: np.asarray(v) for k, v in rec.items()}
    out["maps"] = np.stack(maps).astype(np.int8)
    out["obs_cast_err"] = np.float64(cast_err)
    for k, v in dict(xs=env.xs, ys=env.ys, n_pursuers=P, n_evaders=E, obs_range=env.obs_range, n_catch=env.n_catch,
                     surround=int(env.surround), flatten=int(flatten), include_id=int(env.include_id),
                     reward_global=int(env.reward_mech == "global"), sample_maps=0, train_pursuit=0).items():
        out["cfg_" + k] = np.int64(v)
    for k in ("catchr", "term_pursuit", "urgency_reward", "layer_norm"):
        out["cfg_" + k] = np.float64(getattr(env, k))
    path = os.path.join(OUT, "evadercontrol_%s.npz" % name)
    np.savez_compressed(path, **out)
    print("%-24s ops=%4d steps=%4d removed=%3d none-entries=%4d  %5.1f KB cast_err=%.1e" % (
        name, len(out["op"]), int((out["op"] == 1).sum()), int(out["removed"].sum()), int(out["obs_none"].sum()),
        os.path.getsize(path) / 1024.0, cast_err))


def main():
    R = ref_loader.load()
    TM = R["TwoDMaps"]
    run(R, "surround_local", [TM.rectangle_map(16, 16)],
        dict(n_evaders=12, n_pursuers=8, obs_range=7, n_catch=2, surround=True, flatten=True, reward_mech="local"),
        episodes=6, steps=50, seed=21)
    run(R, "colocate_global_hwc", [TM.rectangle_map(10, 10)],
        dict(n_evaders=6, n_pursuers=5, obs_range=5, n_catch=2, surround=False, flatten=False, reward_mech="global",
             catchr=0.1, urgency_reward=-0.05), episodes=6, steps=40, seed=22)


if __name__ == "__main__":
    main()
