#!/usr/bin/env python
"""Generate tests/golden/multiwalker_box2d_*.npz from the UNMODIFIED reference MultiWalkerEnv (multi_walker.py) -- wherever pybox2d
can be imported.  TEST INFRASTRUCTURE ONLY.

The build image has no Box2D (neither the Python package nor a C++ tree to compile), so in this repository the script has never
produced a file and the MultiWalker oracles stay PARITY UNPINNED.  It is committed so that anybody with `pip install
box2d-py` and the reference checkout can pin them:

    MADRL_REFERENCE_ROOT=/path/to/MADRL python oracle/make_golden_multiwalker.py
    python -m pytest tests/test_oracle_multiwalker_golden.py        # skips while tests/golden/multiwalker_box2d_*.npz do not exist

Exit status 0 and a message when Box2D is missing (nothing written).

    python oracle/make_golden_multiwalker.py --envlayer      # what this repository CAN record: tests/golden/multiwalker_envlayer_*.npz

--envlayer puts oracle/shims_box2d on the path: a package named `Box2D` whose b2World is the World of oracle/multiwalker_ref.c (the
independent restatement of the Box2D 2.3.0 subset the env uses).  The UNMODIFIED reference module then imports and runs here: its reset()
builds the world call by call, its apply_action / get_observation / ContactDetector / LidarCallback / rewards / termination run on the restated
dynamics.  Those files pin this repository's ENV LAYER (and the world its reset constructs) to the reference's own code --
tests/test_multiwalker_envlayer.py replays them through the oracle, the product source and the kernels, free-running -- and say nothing
about the dynamics (b2World::Step stays restated: PARITY UNPINNED).

What is recorded, per episode of n_walkers walkers under seeded random actions with stretches of zero actions (walkers collapse:
hull contacts, dropped package, resets):
  terrain_y [NT] float64        env.terrain_y after reset (multi_walker.py:516-612)            -> reset_with(terrain=)
  push      [W]  float64        the initial hull pushes: np_random.uniform(-5, 5) per walker (:130-131), captured from the RNG
  actions   [T, W, 4] float32
  bodies    [T + 1, NB, 6] float32   every body's (worldCenter.x, .y, angle, linearVelocity.x, .y, angularVelocity) BEFORE step t
                                (row 0: after reset) in the order package, then per walker hull, leg0 upper, leg0 lower, leg1 upper,
                                leg1 lower -- what the oracles are teacher-forced on
  obs       [T + 1, W, 32] float32   reset() / step() observations (position / angle noise set to 0)
  rew       [T, W] float64 ; done [T] uint8
  flags     [T + 1, 1 + 3W] uint8    game_over, fallen[W], ground_contact[W][2] (ContactDetector :50-84)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def bodies_of(env):
    rows = []
    def rec(b):
        c = b.worldCenter
        rows.append([c[0], c[1], b.angle, b.linearVelocity[0], b.linearVelocity[1], b.angularVelocity])
    rec(env.package)
    for w in env.walkers:
        rec(w.hull)
        for leg in w.legs:      # [upper0, lower0, upper1, lower1] (:136-179)
            rec(leg)
    return np.asarray(rows, np.float32)


def flags_of(env):
    f = [int(bool(env.game_over))] + [int(bool(x)) for x in env.fallen_walkers]
    for w in env.walkers:
        f += [int(bool(w.legs[1].ground_contact)), int(bool(w.legs[3].ground_contact))]
    return np.asarray(f, np.uint8)


def run(MultiWalkerEnv, name, n_walkers, reward_mech, episodes, steps, seed, prefix="multiwalker_box2d_", zero_from=44, **env_kw):
    rng = np.random.RandomState(seed)
    out = []
    for ep in range(episodes):
        env = MultiWalkerEnv(n_walkers=n_walkers, position_noise=0.0, angle_noise=0.0, reward_mech=reward_mech, **env_kw)
        env.seed(int(rng.randint(2 ** 31 - 1)))
        pushes = []
        real_uniform = env.np_random.uniform
        def spy(lo, hi, *a, **k):
            v = real_uniform(lo, hi, *a, **k)
            if np.isscalar(v) and (lo, hi) == (-5.0, 5.0):     # INITIAL_RANDOM (:130)
                pushes.append(float(v))
            return v
        env.np_random.uniform = spy
        obs0 = env.reset()
        env.np_random.uniform = real_uniform
        rec = dict(terrain_y=np.asarray(env.terrain_y, np.float64), push=np.asarray(pushes[-n_walkers:], np.float64), actions=[], bodies=[bodies_of(env)],
                   obs=[np.asarray(obs0, np.float32)], rew=[], done=[], flags=[flags_of(env)])
        for t in range(steps):
            a = rng.uniform(-1, 1, (n_walkers, 4)).astype(np.float32)
            if t % 60 > zero_from:
                a[:] = 0
            o, r, d, _ = env.step(a)
            rec["actions"].append(a); rec["obs"].append(np.asarray(o, np.float32)); rec["rew"].append(np.asarray(r, np.float64).reshape(-1) * np.ones(n_walkers))
            rec["done"].append(int(bool(d))); rec["bodies"].append(bodies_of(env)); rec["flags"].append(flags_of(env))
            if d:
                break
        out.append({k: np.asarray(v) for k, v in rec.items()})
    path = os.path.join(OUT, "%s%s.npz" % (prefix, name))
    flat = {"n_episodes": np.int64(len(out)), "n_walkers": np.int64(n_walkers), "reward_global": np.int64(reward_mech == "global")}
    for k in ("forward_reward", "fall_reward", "drop_reward"):
        flat["cfg_" + k] = np.float64(getattr(env, k))
    flat["cfg_terminate_on_fall"] = np.int64(bool(env.terminate_on_fall))
    for i, r in enumerate(out):
        for k, v in r.items():
            flat["ep%d_%s" % (i, k)] = v
    np.savez_compressed(path, **flat)
    print("%s: %d episodes, %d steps, %.1f KB" % (path, len(out), sum(len(r["done"]) for r in out), os.path.getsize(path) / 1024.0))


def main():
    envlayer = "--envlayer" in sys.argv[1:]
    root = os.environ.get("MADRL_REFERENCE_ROOT", "/root/reference")
    if envlayer:
        sys.path.insert(0, os.path.join(HERE, "shims_box2d"))
    try:
        import Box2D
    except Exception as e:  # pragma: no cover
        print("Box2D cannot be imported here (%s): no MultiWalker goldens written; the MultiWalker oracles stay PARITY UNPINNED." % e)
        return 0
    if envlayer != ("shim" in getattr(Box2D, "__version__", "")):
        print("refusing to mix the two kinds of recording: --envlayer=%s but `import Box2D` gave %r" % (envlayer, getattr(Box2D, "__version__", "?")))
        return 1
    for p in (root, os.path.join(HERE, "shims")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.setdefault("MPLBACKEND", "Agg")
    from madrl_environments.walker.multi_walker import MultiWalkerEnv
    if not envlayer:
        run(MultiWalkerEnv, "w3_local", 3, "local", episodes=6, steps=200, seed=31)
        run(MultiWalkerEnv, "w2_global", 2, "global", episodes=4, steps=200, seed=32)
        return 0
    # the reference's env layer on the restated dynamics: BASELINE configs[3] (three walkers), the module's own example (two), one and
    # four walkers, both reward mechanisms, every reward coefficient changed, terminate_on_fall off (walkers keep falling, the package is dropped)
    pre = "multiwalker_envlayer_"
    run(MultiWalkerEnv, "w3_local", 3, "local", episodes=6, steps=160, seed=31, prefix=pre)
    run(MultiWalkerEnv, "w2_global", 2, "global", episodes=4, steps=160, seed=32, prefix=pre)
    run(MultiWalkerEnv, "w3_noterminate", 3, "local", episodes=3, steps=200, seed=33, prefix=pre, zero_from=25, terminate_on_fall=False, forward_reward=2.0,
        fall_reward=-7.0, drop_reward=-33.0)
    run(MultiWalkerEnv, "w1_local", 1, "local", episodes=3, steps=120, seed=34, prefix=pre)
    run(MultiWalkerEnv, "w4_global", 4, "global", episodes=3, steps=120, seed=35, prefix=pre, fall_reward=-10.0)
    return 0


if __name__ == "__main__":
    sys.exit(main())
