#!/usr/bin/env python
"""Generate tests/golden/pursuit_*.npz by running the UNMODIFIED reference
PursuitEvade (/root/reference/madrl_environments/pursuit/pursuit_evade.py) under
the shims in oracle/shims.

TEST INFRASTRUCTURE ONLY.  Runs in the build container (the reference tree is
not present on the GPU box); its outputs are committed.

How randomness is pinned (the reference draws from unseeded MT19937 streams that
no GPU kernel can reproduce):
  * initial positions: `agent_utils.feasible_position` (agent_utils.py:31-47) is
    replaced *at run time* by a replay of a recorded position list, so reset()
    (pursuit_evade.py:173-207) still runs unmodified, including its obs.
  * evader moves: `evader_controller=` kwarg (pursuit_evade.py:87) gets a scripted
    controller; one act() per REMAINING evader in layer order (:238-241).
  * sample_maps: np.random is seeded, the chosen map is recorded by identity.

Each file holds a sequence of ops on ONE env object (so stale-observation state,
Q2, carries across resets):  op[t]==0 -> reset(), op[t]==1 -> step(actions).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_loader  # noqa: E402

OUT = os.path.join(os.environ.get("MADRL_GOLDEN_OUT") or os.path.join(os.path.dirname(HERE), "tests", "golden"))


class ScriptedController(object):
    """Replays evader actions: queue refilled once per step by the driver."""

    def __init__(self):
        self.queue = []
        self.used = 0

    def load(self, actions):
        self.queue = list(actions)
        self.used = 0

    def act(self, state):
        a = self.queue[self.used]
        self.used += 1
        return int(a)


def free_cells(m):
    return np.argwhere(m != -1)


def run_scenario(R, name, maps, cfg, episodes, steps_per_episode, seed, in_building=False,
                 chase=0.5, extra_after_done=3):
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        return   # `make_golden_pursuit.py <name> ...` regenerates only the named scenarios
    PursuitEvade = R["PursuitEvade"]
    from madrl_environments.pursuit.utils import agent_utils

    rng = np.random.RandomState(seed)
    maps = [np.asarray(m, dtype=np.int32) for m in maps]
    ctrl = ScriptedController()
    kwargs = dict(cfg)
    kwargs["evader_controller"] = ctrl
    env = PursuitEvade(maps, **kwargs)
    if kwargs.get("random_opponents"):
        env.seed(seed)   # the evader count comes from the env's own generator
    P, E = env.n_pursuers, env.n_evaders
    xs, ys = env.xs, env.ys
    flatten = kwargs.get("flatten", True)

    pos_queue = []

    def replay_feasible_position(map_matrix, constraints=None):
        return pos_queue.pop(0)

    orig_fp = agent_utils.feasible_position
    agent_utils.feasible_position = replay_feasible_position

    rec = dict(op=[], map_id=[], init_p=[], init_e=[], act_p=[], act_e=[], obs=[], rew=[],
               done=[], removed=[], pos_p=[], pos_e=[], gone_e=[], n_ev_draws=[])

    def snapshot():
        pp = np.array([env.pursuer_layer.get_position(i).copy() for i in range(P)], dtype=np.int32)
        pe = -np.ones((E, 2), dtype=np.int32)
        k = 0
        gone = env.evaders_gone.astype(np.uint8).copy()
        for i in range(E):
            if i >= env.n_evaders:      # random_opponents: this episode has fewer evaders; absent slots count as gone
                gone[i] = 1
            elif not env.evaders_gone[i]:
                pe[i] = env.evader_layer.get_position(k)
                k += 1
        assert k == env.evader_layer.n_agents()
        return pp, pe, gone

    def obs_array(obslist):
        if flatten:
            return np.stack([np.asarray(o, dtype=np.float64) for o in obslist])
        # non-flatten obs are (R,R,4) VIEWS into local_obs (Q3): copy now
        return np.stack([np.array(o, dtype=np.float64, copy=True) for o in obslist])

    try:
        for ep in range(episodes):
            # ---- reset with injected positions
            np.random.seed(rng.randint(2**31 - 1))
            st = np.random.get_state()
            if env.sample_maps:
                # peek which map reset() will draw (pursuit_evade.py:183 is the first draw)
                mid = int(np.random.randint(len(maps)))
                np.random.set_state(st)
            else:
                mid = 0
            m = maps[mid]
            cells = free_cells(m) if not in_building else np.argwhere(np.ones_like(m) > 0)
            ip = cells[rng.randint(len(cells), size=P)]
            ie = cells[rng.randint(len(cells), size=E)]
            if ep % 2 == 1 and not in_building:
                # cluster pursuers near evaders so that catches happen
                ie = cells[rng.randint(min(len(cells), 6), size=E)]
                ip = cells[rng.randint(min(len(cells), 8), size=P)]
            n_ev = E
            if getattr(env, "random_opponents", False):
                # peek the evader count reset() will draw from the env's own generator (pursuit_evade.py:177-181)
                st2 = env.np_random.get_state()
                n_ev = int(env.np_random.randint(1, env.max_opponents))
                env.np_random.set_state(st2)
                assert n_ev <= E
                ie = ie.copy(); ie[n_ev:] = -1      # slots that are not created this episode
            pos_queue[:] = [tuple(int(v) for v in p) for p in ip] + [tuple(int(v) for v in p) for p in ie[:n_ev]]
            obs = env.reset()
            assert env.n_evaders == n_ev
            assert not pos_queue
            assert env.map_matrix is maps[mid]
            pp, pe, ge = snapshot()
            rec["op"].append(0); rec["map_id"].append(mid)
            rec["init_p"].append(ip.astype(np.int32)); rec["init_e"].append(ie.astype(np.int32))
            rec["act_p"].append(np.zeros(P, np.int32)); rec["act_e"].append(np.full(E, 4, np.int32))
            rec["obs"].append(obs_array(obs)); rec["rew"].append(np.zeros(P))
            rec["done"].append(0); rec["removed"].append(0)
            rec["pos_p"].append(pp); rec["pos_e"].append(pe); rec["gone_e"].append(ge)
            rec["n_ev_draws"].append(0)
            after_done = 0
            for t in range(steps_per_episode):
                # pursuer actions: random, biased to chase the nearest remaining evader
                ap = rng.randint(5, size=P).astype(np.int32)
                alive = [i for i in range(env.n_evaders) if not env.evaders_gone[i]]
                if alive:
                    for j in range(P):
                        if rng.rand() < chase:
                            d = pe[alive] - pp[j]
                            k = np.argmin(np.abs(d).sum(1))
                            dx, dy = d[k]
                            if abs(dx) + abs(dy) <= 1:
                                ap[j] = 4
                            elif abs(dx) >= abs(dy):
                                ap[j] = 1 if dx > 0 else 0
                            else:
                                ap[j] = 2 if dy > 0 else 3
                n_alive = env.evader_layer.n_agents()
                ae = np.full(E, 4, np.int32)
                # evaders mostly stay put in odd episodes so they get surrounded
                if ep % 2 == 1:
                    ae[:n_alive] = np.where(rng.rand(n_alive) < 0.7, 4, rng.randint(5, size=n_alive))
                else:
                    ae[:n_alive] = rng.randint(5, size=n_alive)
                ctrl.load(ae[:n_alive])
                obs, rew, done, info = env.step(ap.copy())
                assert ctrl.used == n_alive
                pp, pe, ge = snapshot()
                rec["op"].append(1); rec["map_id"].append(mid)
                rec["init_p"].append(np.zeros((P, 2), np.int32)); rec["init_e"].append(np.zeros((E, 2), np.int32))
                rec["act_p"].append(ap); rec["act_e"].append(ae)
                rec["obs"].append(obs_array(obs)); rec["rew"].append(np.asarray(rew, dtype=np.float64))
                rec["done"].append(int(bool(done))); rec["removed"].append(int(info["removed"]))
                rec["pos_p"].append(pp); rec["pos_e"].append(pe); rec["gone_e"].append(ge)
                rec["n_ev_draws"].append(n_alive)
                if done:
                    after_done += 1
                    if after_done > extra_after_done:
                        break
    finally:
        agent_utils.feasible_position = orig_fp

    out = {k: np.asarray(v) for k, v in rec.items()}
    obs64 = out.pop("obs")
    out["obs_f32"] = obs64.astype(np.float32)
    # every observation value must survive the f64->f32->f64 round trip up to the
    # representation of k/10; record the max cast error as evidence.
    out["obs_cast_err"] = np.float64(np.abs(obs64 - out["obs_f32"].astype(np.float64)).max())
    out["rew_f64"] = out.pop("rew")
    out["maps"] = np.stack(maps).astype(np.int8)
    scalars = dict(xs=xs, ys=ys, n_pursuers=P, n_evaders=E, obs_range=env.obs_range,
                   random_opponents=int(getattr(env, "random_opponents", False)), max_opponents=int(getattr(env, "max_opponents", 10)),
                   n_catch=env.n_catch, surround=int(env.surround), flatten=int(flatten),
                   include_id=int(env.include_id), reward_global=int(env.reward_mech == "global"),
                   sample_maps=int(env.sample_maps))
    for k, v in scalars.items():
        out["cfg_" + k] = np.int64(v)
    for k in ("catchr", "term_pursuit", "urgency_reward", "layer_norm"):
        out["cfg_" + k] = np.float64(getattr(env, k))
    path = os.path.join(OUT, "pursuit_%s.npz" % name)
    np.savez_compressed(path, **out)
    nsteps = int((out["op"] == 1).sum())
    print("%-28s ops=%4d steps=%4d removed=%3d dones=%3d  %6.1f KB  cast_err=%.2e" % (
        name, len(out["op"]), nsteps, int(out["removed"].sum()), int(out["done"].sum()),
        os.path.getsize(path) / 1024.0, float(out["obs_cast_err"])))


def main():
    R = ref_loader.load()
    TM = R["TwoDMaps"]
    os.makedirs(OUT, exist_ok=True)
    pool16 = np.load(os.path.join(ref_loader.REFERENCE_ROOT, "maps", "map_pool16.npy"))
    rect16 = TM.rectangle_map(16, 16)
    rect32 = TM.rectangle_map(32, 32)

    # A: BASELINE C1/C2 configuration (SURVEY 8(d) "C1 input")
    run_scenario(R, "c1_surround_local", [rect16],
                 dict(n_evaders=30, n_pursuers=8, obs_range=7, n_catch=2, surround=True,
                      flatten=True, reward_mech="local"), episodes=4, steps_per_episode=60, seed=1)
    # B: runner defaults for rewards (run_pursuit.py:27-28), global reward mean
    run_scenario(R, "c1_surround_global", [rect16],
                 dict(n_evaders=30, n_pursuers=8, obs_range=7, n_catch=2, surround=True,
                      flatten=True, reward_mech="global", catchr=0.1, term_pursuit=5.0,
                      urgency_reward=-0.1), episodes=4, steps_per_episode=50, seed=2)
    # C: heuristics/pursuit.py:66-67 family: co-location catch, (R,R,4) observations
    run_scenario(R, "c1_colocate_hwc", [rect16],
                 dict(n_evaders=30, n_pursuers=8, obs_range=7, n_catch=2, surround=False,
                      flatten=False, reward_mech="local"), episodes=4, steps_per_episode=50, seed=3)
    # D: map pool with walls in row/col 0 (Q4), sample_maps
    run_scenario(R, "pool16_sample_maps", list(pool16),
                 dict(n_evaders=30, n_pursuers=8, obs_range=7, n_catch=2, surround=True,
                      flatten=True, reward_mech="local", sample_maps=True),
                 episodes=8, steps_per_episode=40, seed=4)
    # E: tiny dense world, episodes run to done and beyond
    run_scenario(R, "tiny5_dense", [np.zeros((5, 5), np.int32)],
                 dict(n_evaders=3, n_pursuers=4, obs_range=3, n_catch=2, surround=True,
                      flatten=True, reward_mech="local"), episodes=8, steps_per_episode=60, seed=5,
                 chase=0.9)
    # F: BASELINE C5 configuration
    run_scenario(R, "c5_32x32", [rect32],
                 dict(n_evaders=60, n_pursuers=16, obs_range=7, n_catch=2, surround=True,
                      flatten=True, reward_mech="local"), episodes=2, steps_per_episode=40, seed=6)
    # G: even obs_range (Q11), no id, global mean over 11 pursuers (numpy pairwise tail)
    run_scenario(R, "even_range_noid", [TM.rectangle_map(10, 10)],
                 dict(n_evaders=9, n_pursuers=11, obs_range=4, n_catch=2, surround=True,
                      flatten=True, include_id=False, reward_mech="global", urgency_reward=-0.05),
                 episodes=4, steps_per_episode=40, seed=7)
    # H: agents dropped inside buildings become terminal (DiscreteAgent.py:75-78)
    run_scenario(R, "in_building", [TM.rectangle_map(8, 8)],
                 dict(n_evaders=6, n_pursuers=5, obs_range=5, n_catch=1, surround=False,
                      flatten=True, reward_mech="local"), episodes=4, steps_per_episode=30, seed=8,
                 in_building=True)
    # I: non-square map (x/y transposition check), co-location n_catch=1
    run_scenario(R, "nonsquare_12x20", [TM.rectangle_map(12, 20)],
                 dict(n_evaders=12, n_pursuers=6, obs_range=5, n_catch=1, surround=False,
                      flatten=True, reward_mech="local", catchr=0.1), episodes=4,
                 steps_per_episode=40, seed=9)
    # J: window wider than the map (mostly out-of-bounds cells)
    run_scenario(R, "window_gt_map", [np.zeros((6, 6), np.int32)],
                 dict(n_evaders=5, n_pursuers=3, obs_range=11, n_catch=2, surround=True,
                      flatten=False, reward_mech="global"), episodes=4, steps_per_episode=40,
                 seed=10, chase=0.8)
    # K: more agents than one wavefront has lanes, bigger window
    run_scenario(R, "wide_70v90", [TM.rectangle_map(24, 24)],
                 dict(n_evaders=90, n_pursuers=70, obs_range=9, n_catch=3, surround=True,
                      flatten=True, reward_mech="global"), episodes=2, steps_per_episode=25,
                 seed=11)
    # L: random_opponents (train_pursuit): the number of evaders is redrawn by every reset (pursuit_evade.py:177-181);
    #    n_evaders >= max_opponents - 1, otherwise the reference indexes past evaders_gone (:138, :467)
    run_scenario(R, "random_opponents", [rect16],
                 dict(n_evaders=9, n_pursuers=5, obs_range=5, n_catch=2, surround=True, flatten=True,
                      reward_mech="local", random_opponents=True, max_opponents=10), episodes=10,
                 steps_per_episode=30, seed=12, chase=0.8)
    # M: more evaders than a byte-sized agent count (the authors' largest launch line runs 100 pursuers / 300 evaders,
    #    runners/old/rllab/pursuit_cnn.sh:1): a crowded 24 x 24 map
    run_scenario(R, "crowd_20v300", [TM.rectangle_map(24, 24)],
                 dict(n_evaders=300, n_pursuers=20, obs_range=9, n_catch=2, surround=True, flatten=True,
                      reward_mech="local"), episodes=2, steps_per_episode=14, seed=13)
    # N: more pursuers than numpy's pairwise sum takes in one block (rewards.mean() over 260 values splits twice, :261)
    run_scenario(R, "crowd_260v40_global", [TM.rectangle_map(20, 20)],
                 dict(n_evaders=40, n_pursuers=260, obs_range=5, n_catch=3, surround=False, flatten=True,
                      reward_mech="global", catchr=0.1, urgency_reward=-0.05), episodes=2, steps_per_episode=10, seed=14)
    # O, P: the authors' own training shapes -- runners/old/rllab/pursuit.sh:1 (30 pursuers / 50 evaders, obs_range 11, --sample_maps
    #    --flatten --surround, local reward) and runners/old/rltools/pursuit.sh:1 (30 v 30, --catchr 0.0 --term_pursuit 5.0).  Their
    #    map_pool32.npy is not in the tree: the survey's substitute, TwoDMaps.resize(2, map_pool16) (32 x 32, walls in row / column 0)
    pool32 = list(TM.resize(2, pool16))
    run_scenario(R, "authors_30v50_obs11", pool32,
                 dict(n_evaders=50, n_pursuers=30, obs_range=11, n_catch=2, surround=True, flatten=True, reward_mech="local",
                      sample_maps=True), episodes=4, steps_per_episode=24, seed=15, chase=0.8)
    run_scenario(R, "authors_30v30_obs11", pool32,
                 dict(n_evaders=30, n_pursuers=30, obs_range=11, n_catch=2, surround=True, flatten=True, reward_mech="local",
                      sample_maps=True, catchr=0.0, term_pursuit=5.0), episodes=4, steps_per_episode=24, seed=16, chase=0.8)


if __name__ == "__main__":
    main()
