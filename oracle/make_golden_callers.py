#!/usr/bin/env python
"""Golden records of the reference's own CALLERS driving the reference PursuitEvade -- what a user of the drop-in
N == 1 env runs unchanged.  TEST INFRASTRUCTURE ONLY; runs in the build container, the output
tests/golden/callers_pursuit.npz is committed and replayed on the GPU by tests/test_dropin_callers_gpu.py.

Callers recorded (all UNMODIFIED reference code, imported from /root/reference):
  A  AbstractMAEnv.animate(act_fn, nsteps)                         madrl_environments/__init__.py:72-107
     (render() replaced on the INSTANCE by a no-op: matplotlib is host-side decoration)
  B  DiagnosticsWrapper(StandardizedEnv(env, ...)).reset()/step()  madrl_environments/__init__.py:204-311, :314-369
  C  the rollout loop of heuristics/pursuit.py:64-85 with PursuitHeuristicPolicy (heuristics/pursuit.py:13-56)
  (ObservationBuffer cannot be recorded this way: its constructor assigns to its own read-only `reward_mech` property,
   madrl_environments/__init__.py:151, and raises AttributeError on any env; tests/golden/wrappers_replay.npz covers its
   arithmetic.)

Randomness is pinned exactly as in make_golden_pursuit.py: reset positions replayed through
agent_utils.feasible_position, evader moves through the `evader_controller=` kwarg (one act() per remaining evader).
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_loader  # noqa: E402
from oracle.make_golden_heuristics import Py2Array  # noqa: E402

OUT = os.path.join(os.environ.get("MADRL_GOLDEN_OUT") or os.path.join(os.path.dirname(HERE), "tests", "golden"), "callers_pursuit.npz")


class QueueController(object):
    """evader_controller: answers from one flat pre-drawn list, whatever the number of remaining evaders"""

    def __init__(self, actions):
        self.actions, self.used = np.asarray(actions), 0

    def act(self, state):
        a = int(self.actions[self.used])
        self.used += 1
        return a


def act_fn(o):
    """deterministic function of one agent's observation (identical values on both sides -> identical actions)"""
    return int(np.floor(np.sum(np.asarray(o, dtype=np.float64)) * 7.0)) % 5


class SeededSpace(object):
    def __init__(self, seed):
        self.rng = np.random.RandomState(seed)

    def sample(self):
        return int(self.rng.randint(5))


def main():
    R = ref_loader.load()
    from madrl_environments.pursuit.utils import agent_utils
    import madrl_environments as me
    PursuitEvade, TM = R["PursuitEvade"], R["TwoDMaps"]
    rect = TM.rectangle_map(16, 16)
    free = np.argwhere(rect != -1)
    rng = np.random.RandomState(11)
    out = {}
    pos_queue = []
    orig_fp = agent_utils.feasible_position
    agent_utils.feasible_position = lambda map_matrix, constraints=None: pos_queue.pop(0)

    def positions(n_resets, P, E, clustered=False):
        res = []
        for _ in range(n_resets):
            if clustered:
                c = free[rng.randint(len(free))]
                near = free[np.abs(free - c).sum(1) <= 4]
                res.append(near[rng.randint(len(near), size=P + E)].astype(np.int32))
            else:
                res.append(free[rng.randint(len(free), size=P + E)].astype(np.int32))
        return np.stack(res)

    def queue(pos):
        pos_queue.extend(tuple(int(v) for v in p) for p in pos)

    try:
        # ------------------------------------------------------------------ A: animate
        P, E = 8, 30
        kw = dict(n_evaders=E, n_pursuers=P, obs_range=7, n_catch=2, surround=True, flatten=True, reward_mech="local")
        eacts = rng.randint(5, size=60 * E)
        eacts[rng.rand(len(eacts)) < 0.5] = 4   # evaders often stay: some get surrounded
        pos = positions(1, P, E, clustered=True)
        env = PursuitEvade([rect], evader_controller=QueueController(eacts), **kw)
        env.render = lambda *a, **k: None
        seen = []

        def logging_act_fn(o):
            seen.append(np.array(o, dtype=np.float64))
            return act_fn(o)

        queue(pos[0])
        # PursuitEvade overrides animate() with an mp4 writer (pursuit_evade.py:297-326: save_image + ffmpeg); the generic
        # loop every other env inherits is AbstractMAEnv.animate, called here on the reference env explicitly
        rew, traj_info = me.AbstractMAEnv.animate(env, logging_act_fn, 60)
        out.update(a_pos=pos, a_eacts=eacts, a_rew=np.asarray(rew, dtype=np.float64), a_removed=np.asarray(traj_info["removed"]),
                   a_obs=np.stack(seen).reshape(-1, P, seen[0].shape[0]).astype(np.float32), a_nsteps=np.int64(60))
        print("A animate: %d steps, rew sum %.3f, removed %d" % (len(traj_info["removed"]), float(np.sum(rew)), int(np.sum(traj_info["removed"]))))

        # ------------------------------------------------------------------ B: DiagnosticsWrapper(StandardizedEnv(env))
        T = 70
        eacts = rng.randint(5, size=T * E)
        pacts = rng.randint(5, size=(T, P)).astype(np.int32)
        pos = positions(6, P, E)
        env = PursuitEvade([rect], evader_controller=QueueController(eacts), **kw)
        cfg = dict(scale_reward=0.5, enable_obsnorm=True, enable_rewnorm=True, obs_alpha=0.01, rew_alpha=0.01, eps=1e-8)
        w = me.DiagnosticsWrapper(me.StandardizedEnv(env, **cfg), discount=0.95, max_traj_len=25)
        n_reset = 0
        queue(pos[n_reset]); n_reset += 1
        b_obs, b_rew, b_done, b_op, b_log = [np.stack(w.reset())], [np.zeros(P)], [0], [0], [np.full(4, np.nan)]
        for t in range(T):
            o, r, d, log = w.step(pacts[t])
            b_obs.append(np.stack(o)); b_rew.append(np.asarray(r, dtype=np.float64)); b_done.append(int(bool(d))); b_op.append(1)
            if "global/episode_length" in log:
                b_log.append(np.array([log["global/episode_avg_reward"], log["global/episode_disc_return"], log["global/episode_length"],
                                       log["global/episode_reward_agent3"]], dtype=np.float64))
                queue(pos[n_reset]); n_reset += 1
                b_obs.append(np.stack(w.reset())); b_rew.append(np.zeros(P)); b_done.append(0); b_op.append(0); b_log.append(np.full(4, np.nan))
            else:
                b_log.append(np.full(4, np.nan))
        out.update(b_pos=pos[:n_reset], b_eacts=eacts, b_pacts=pacts, b_obs=np.stack(b_obs), b_rew=np.stack(b_rew), b_done=np.asarray(b_done),
                   b_op=np.asarray(b_op), b_log=np.stack(b_log), **{"b_cfg_" + k: np.float64(v) for k, v in cfg.items()},
                   b_discount=np.float64(0.95), b_max_traj_len=np.int64(25))
        print("B wrappers: %d ops, %d episodes closed" % (len(b_op), n_reset - 1))

        # ------------------------------------------------------------------ C: heuristics/pursuit.py:64-85
        hp = importlib.import_module("heuristics.pursuit")
        kwc = dict(n_evaders=30, n_pursuers=8, obs_range=7, n_catch=4, surround=False, flatten=False)  # heuristics/pursuit.py:64-65
        T = 120
        eacts = rng.randint(5, size=T * 30)
        eacts[rng.rand(len(eacts)) < 0.6] = 4
        pos = positions(1, 8, 30)
        env = PursuitEvade([rect], evader_controller=QueueController(eacts), **kwc)
        policy = hp.PursuitHeuristicPolicy(env.agents[0].observation_space, SeededSpace(5))
        queue(pos[0])
        obs = env.reset()
        c_act, c_obs, c_rew, c_done, c_removed, total = [], [np.stack([np.array(o) for o in obs])], [], [], [], 0.0
        for _ in range(T):
            act_list = []
            for o in obs:
                a, _ = policy.sample_actions(np.asarray(o).view(Py2Array))
                act_list.append(a)
            obs, r, done, info = env.step(act_list)
            total += np.mean(r)
            c_act.append(np.asarray(act_list, dtype=np.int32)); c_obs.append(np.stack([np.array(o) for o in obs]))
            c_rew.append(np.asarray(r, dtype=np.float64)); c_done.append(int(bool(done))); c_removed.append(int(info["removed"]))
            if done:
                break
        out.update(c_pos=pos, c_eacts=eacts, c_act=np.stack(c_act), c_obs=np.stack(c_obs).astype(np.float32), c_rew=np.stack(c_rew),
                   c_done=np.asarray(c_done), c_removed=np.asarray(c_removed), c_total=np.float64(total))
        print("C heuristic loop: %d steps, removed %d, mean-reward sum %.3f" % (len(c_act), int(np.sum(c_removed)), total))

    finally:
        agent_utils.feasible_position = orig_fp
    out["map"] = np.asarray(rect, dtype=np.int8)
    np.savez_compressed(OUT, **out)
    print("wrote %s (%.1f KB)" % (OUT, os.path.getsize(OUT) / 1024.0))


if __name__ == "__main__":
    main()
