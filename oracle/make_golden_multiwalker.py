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
  obs       [T + 1, W, 32] float64   reset() / step() observations (position / angle noise set to 0)
  rew       [T, W] float64 ; done [T] uint8
  flags     [T + 1, 1 + 3W] uint8    game_over, fallen[W], ground_contact[W][2] (ContactDetector :50-84)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.environ.get("MADRL_GOLDEN_OUT") or os.path.join(os.path.dirname(HERE), "tests", "golden"))


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


class ScriptedNormal(object):
    """np.random.normal for the observation noise (:389-395; the reference draws it from the global, unseeded generator) replaced, while a
    recording runs, by the draws THIS repository's RNG contract assigns to (seed, env id, episode, observation): Philox4x32-10 with counter
    (env id, episode, observation << 6 | 4 i + q, tag 34), Box-Muller on (r.x, r.y) -> z[2q], z[2q + 1]; walker i uses z[0..3] for the
    neighbours it has, in order, and z[4], z[5], z[6] for the package.  The reference asks for loc + scale * z call by call; which z goes
    with which call is mirrored here from the ORDER of its calls only."""

    def __init__(self, seed, gid, n_walkers):
        self.key, self.gid, self.W = [seed & 0xFFFFFFFF, seed >> 32], gid, n_walkers
        self.episode, self.tick, self.queue = 1, 0, []

    def _fill(self):
        import math
        from oracle import pursuit as po
        for i in range(self.W):
            z = []
            for q in range(4):
                r = po.philox([self.gid, self.episode, (self.tick << 6) | (4 * i + q), 34], self.key)
                u1, u2 = float((int(r[0]) >> 8) + 1) / 16777216.0, float(int(r[1]) >> 8) / 16777216.0
                rad = math.sqrt(-2.0 * math.log(u1))
                z += [rad * math.cos(2.0 * 3.14159265358979323846 * u2), rad * math.sin(2.0 * 3.14159265358979323846 * u2)]
            n_nb = (i - 1 >= 0) + (i + 1 < self.W)
            self.queue += z[:2 * n_nb] + z[4:7]
        self.tick += 1

    def __call__(self, loc=0.0, scale=1.0, size=None):
        assert size is None
        if not self.queue:
            self._fill()
        return loc + scale * self.queue.pop(0)


N_DRAWN = 12


def run(MultiWalkerEnv, name, n_walkers, reward_mech, episodes, steps, seed, prefix="multiwalker_box2d_", zero_from=44, noise=None, fresh_world=False, **env_kw):
    """noise = (position_noise, angle_noise, seed, first env id): the observation noise on, scripted (ScriptedNormal); episode k is env id + k"""
    rng = np.random.RandomState(seed)
    out = []
    real_normal = np.random.normal
    for ep in range(episodes):
        if noise:
            np.random.normal = ScriptedNormal(noise[2], noise[3] + ep, n_walkers)
        env = MultiWalkerEnv(n_walkers=n_walkers, position_noise=noise[0] if noise else 0.0, angle_noise=noise[1] if noise else 0.0, reward_mech=reward_mech, **env_kw)
        if noise:   # the constructor ran a whole reset() of its own (setup(), :303): the recording starts at the next one, observation 0 of episode 1
            np.random.normal = ScriptedNormal(noise[2], noise[3] + ep, n_walkers)
        env.seed(int(rng.randint(2 ** 31 - 1)))
        pushes = []

        class SpyRandom(object):
            """the walkers' own generators (BipedalWalker._seed, :109-111) draw the initial pushes (:130-131)"""

            def __init__(self, rs):
                self.rs = rs

            def uniform(self, lo, hi, *a, **k):
                v = self.rs.uniform(lo, hi, *a, **k)
                if np.isscalar(v) and (lo, hi) == (-5, 5):     # INITIAL_RANDOM
                    pushes.append(float(v))
                return v

            def __getattr__(self, k):
                return getattr(self.rs, k)

        for w in env.walkers:
            w._seed(int(rng.randint(2 ** 31 - 1)))          # (BipedalWalker seeds itself from the clock otherwise: recordings would not regenerate)
            w.np_random = SpyRandom(w.np_random)
        if fresh_world:
            # The recorded episode is the env object's SECOND reset (the constructor ran the first, :303).  In real Box2D the bodies of a second
            # reset take recycled nodes off the dynamic tree's free list, so the proxy ids -- and with them the order of the contacts created in
            # one FindNewContacts call, which decides bits from the first step on -- depend on the env object's history.  The
            # restatements model the FIRST episode of an env object: give the module a brand-new b2World for the recorded reset, through its own
            # attributes (nothing of the old world is touched: _destroy() returns at once when `terrain` / `hull` are unset, :98-100, :318-319).
            import Box2D
            env.world = Box2D.b2World()
            env.terrain = None
            for w in env.walkers:
                w.world, w.hull = env.world, None
        obs0 = env.reset()
        for w in env.walkers:
            w.np_random = w.np_random.rs
        assert len(pushes) == n_walkers
        rec = dict(terrain_y=np.asarray(env.terrain_y, np.float64), push=np.asarray(pushes[-n_walkers:], np.float64), actions=[], bodies=[bodies_of(env)],
                   obs=[np.asarray(obs0, np.float64)], rew=[], done=[], flags=[flags_of(env)])
        for t in range(steps):
            a = rng.uniform(-1, 1, (n_walkers, 4)).astype(np.float32)
            if t % 60 > zero_from:
                a[:] = 0
            o, r, d, _ = env.step(a)
            rec["actions"].append(a); rec["obs"].append(np.asarray(o, np.float64)); rec["rew"].append(np.asarray(r, np.float64).reshape(-1) * np.ones(n_walkers))
            rec["done"].append(int(bool(d))); rec["bodies"].append(bodies_of(env)); rec["flags"].append(flags_of(env))
            if d:
                break
        if noise:
            assert not np.random.normal.queue, "the reference left noise draws of an observation unused"
        np.random.normal = real_normal
        out.append({k: np.asarray(v) for k, v in rec.items()})
    path = os.path.join(OUT, "%s%s.npz" % (prefix, name))
    flat = {"n_episodes": np.int64(len(out)), "n_walkers": np.int64(n_walkers), "reward_global": np.int64(reward_mech == "global")}
    for k in ("forward_reward", "fall_reward", "drop_reward"):
        flat["cfg_" + k] = np.float64(getattr(env, k))
    flat["cfg_terminate_on_fall"] = np.int64(bool(env.terminate_on_fall))
    flat["cfg_one_hot"] = np.int64(bool(env.one_hot))
    flat["cfg_position_noise"], flat["cfg_angle_noise"] = np.float64(env.position_noise), np.float64(env.angle_noise)
    flat["cfg_seed"], flat["cfg_env_id_base"] = np.uint64(noise[2] if noise else 0), np.int64(noise[3] if noise else 0)
    for i, r in enumerate(out):
        for k, v in r.items():
            flat["ep%d_%s" % (i, k)] = v
    np.savez_compressed(path, **flat)
    print("%s: %d episodes, %d steps, %.1f KB" % (path, len(out), sum(len(r["done"]) for r in out), os.path.getsize(path) / 1024.0))


def record_philox_resets(MultiWalkerEnv, n_walkers, seed, gid0, n, prefix):
    """The reference's reset() fed with the draws THIS repository's reset makes (RNG contract: Philox4x32-10, key = seed,
    counter = (global env id, episode, index, tag 32 terrain | 33 push)): uniform(-1, 1) = 2 u24(r.x) - 1 per terrain point,
    randint(5, 10) = 5 + mulhi(r.y, 5) when the grass counter runs out, push = (2 u24(r.x) - 1) * 5 per walker.  The scripted generator
    hands them to the reference in the order ITS code asks for them; what the reference then builds -- terrain_y (the smoothed walk with
    its one-shot counters, :516-612), the world after reset() and its observation -- is what the plain, un-injected reset of the oracle /
    product / kernels must produce for (seed, env id).  -> <prefix>philox_w<W>.npz"""
    from oracle import pursuit as po
    NT = int(200 * n_walkers / 8)
    key = [seed & 0xFFFFFFFF, seed >> 32]
    u24 = lambda r: float(int(r) >> 8) / 16777216.0
    rec = dict(terrain_y=[], push=[], bodies=[], obs=[], flags=[])
    for gid in range(gid0, gid0 + n):
        script = []
        counter, oneshot = 20, False                          # TERRAIN_STARTPAD; mirrors only WHEN the reference draws, not what it computes
        for i in range(NT):
            r = po.philox([gid, 0, i, 32], key)
            if not oneshot and i > 20:
                script.append(("uniform", (-1, 1), 2.0 * u24(r[0]) - 1.0))
            oneshot = False
            counter -= 1
            if counter == 0:
                counter = 5 + ((int(r[1]) * 5) >> 32)
                script.append(("randint", (5.0, 10), counter))
                oneshot = True
        pushes = [(2.0 * u24(po.philox([gid, 0, w, 33], key)[0]) - 1.0) * 5 for w in range(n_walkers)]

        class Scripted(object):
            def __init__(self, items, rest):
                self.items, self.rest, self.k = items, rest, 0

            def _next(self, kind, args):
                if self.k < len(self.items):
                    k, a, v = self.items[self.k]
                    assert k == kind and tuple(a) == tuple(args), "the reference asked for %s%r where the script holds %s%r" % (kind, args, k, a)
                    self.k += 1
                    return v
                return None

            def uniform(self, lo, hi, *a, **kw):
                v = self._next("uniform", (lo, hi))
                return self.rest.uniform(lo, hi, *a, **kw) if v is None else v

            def randint(self, lo, hi=None, *a, **kw):
                v = self._next("randint", (lo, hi))
                return self.rest.randint(int(lo), hi, *a, **kw) if v is None else v

            def __getattr__(self, k):
                return getattr(self.rest, k)

        env = MultiWalkerEnv(n_walkers=n_walkers, position_noise=0.0, angle_noise=0.0)
        terrain_rng = Scripted(script, np.random.RandomState(1))     # (the clouds, :622-634, draw from the rest)
        env.np_random = terrain_rng
        for w, walker in enumerate(env.walkers):
            walker.np_random = Scripted([("uniform", (-5, 5), pushes[w])], np.random.RandomState(2))
        obs = env.reset()
        assert terrain_rng.k == len(script) and all(wk.np_random.k == 1 for wk in env.walkers), "the reference did not consume the script"
        rec["terrain_y"].append(np.asarray(env.terrain_y, np.float64)); rec["push"].append(np.asarray(pushes, np.float64))
        rec["bodies"].append(bodies_of(env)); rec["obs"].append(np.asarray(obs, np.float64)); rec["flags"].append(flags_of(env))
    path = os.path.join(OUT, "%sphilox_w%d.npz" % (prefix, n_walkers))
    np.savez_compressed(path, n_walkers=np.int64(n_walkers), seed=np.uint64(seed), env_id_base=np.int64(gid0), **{k: np.asarray(v) for k, v in rec.items()})
    print("%s: %d resets, %.1f KB" % (path, n, os.path.getsize(path) / 1024.0))


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
        # real Box2D: every recorded episode on a brand-new b2World (see run(): what the restatements model); one file WITHOUT that, whose episodes
        # are second resets of their env objects -- if it replays too, recycled proxy ids do not matter in practice, if not, that is D1 showing
        run(MultiWalkerEnv, "w3_local", 3, "local", episodes=6, steps=200, seed=31, fresh_world=True)
        run(MultiWalkerEnv, "w2_global", 2, "global", episodes=4, steps=200, seed=32, fresh_world=True)
        run(MultiWalkerEnv, "w10_local", 10, "local", episodes=2, steps=150, seed=41, fresh_world=True, zero_from=30)
        run(MultiWalkerEnv, "w3_second_reset", 3, "local", episodes=3, steps=200, seed=33, fresh_world=False)
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
    run(MultiWalkerEnv, "w2_onehot", 2, "local", episodes=2, steps=60, seed=36, prefix=pre, one_hot=True)   # np.eye(MAX_AGENTS)[i] instead of i / n (:397-400)
    # the reference's DEFAULT configuration has the observation noise on (position_noise = angle_noise = 1e-3, :250)
    run(MultiWalkerEnv, "w3_noise", 3, "local", episodes=4, steps=80, seed=37, prefix=pre, noise=(1e-3, 1e-3, 0xC0FFEE123, 500))
    run(MultiWalkerEnv, "w2_noise_global", 2, "global", episodes=2, steps=60, seed=38, prefix=pre, noise=(5e-3, 2e-2, 11, 0))
    # the rest of the reference's curriculum (lessons/multiwalker/env.yaml: n_walkers 2 .. 10; the package and the terrain grow with the
    # walker count, :293-301): the walker counts the larger capacity classes of the kernels run, default noise on in two of them
    run(MultiWalkerEnv, "w5_local", 5, "local", episodes=3, steps=120, seed=39, prefix=pre)
    run(MultiWalkerEnv, "w8_noise_global", 8, "global", episodes=2, steps=100, seed=40, prefix=pre, noise=(1e-3, 1e-3, 0xABCDEF, 40))
    run(MultiWalkerEnv, "w10_local", 10, "local", episodes=2, steps=120, seed=41, prefix=pre, zero_from=30)
    run(MultiWalkerEnv, "w10_onehot_noise", 10, "local", episodes=2, steps=60, seed=42, prefix=pre, one_hot=True, noise=(1e-3, 1e-3, 99, 7),
        terminate_on_fall=False)
    run(MultiWalkerEnv, "w6_global", 6, "global", episodes=2, steps=100, seed=43, prefix=pre, fall_reward=-20.0)
    run(MultiWalkerEnv, "w7_noterminate", 7, "local", episodes=2, steps=120, seed=44, prefix=pre, zero_from=20, terminate_on_fall=False, drop_reward=-50.0)
    run(MultiWalkerEnv, "w9_local", 9, "local", episodes=2, steps=100, seed=45, prefix=pre)
    # drawn configurations: walker count, reward mechanism, every coefficient, terminate_on_fall, one-hot ids, the observation noise and the step
    # from which the actions are zero all come from one seed -- combinations nobody wrote down get replayed too
    frng = np.random.RandomState(20260927)
    for i in range(N_DRAWN):
        W = int(frng.randint(1, 11))
        kw = dict(forward_reward=float(np.round(frng.uniform(0.5, 3.0), 3)), fall_reward=float(np.round(frng.uniform(-150.0, -1.0), 2)),
                  drop_reward=float(np.round(frng.uniform(-150.0, -1.0), 2)), terminate_on_fall=bool(frng.rand() < 0.6), one_hot=bool(frng.rand() < 0.25))
        noise = (float(10.0 ** frng.uniform(-4, -1.5)), float(10.0 ** frng.uniform(-4, -1.5)), int(frng.randint(1, 2 ** 31 - 1)), int(frng.randint(0, 5000))) if frng.rand() < 0.5 else None
        run(MultiWalkerEnv, "fuzz_%02d" % i, W, "global" if frng.rand() < 0.5 else "local", episodes=2, steps=int(frng.randint(40, 110)), seed=int(frng.randint(1, 2 ** 31 - 1)),
            prefix=pre, zero_from=int(frng.randint(5, 60)), noise=noise, **kw)
    # files named multiwalker_resetdraws_*: a different layout (no episodes), replayed by its own test
    record_philox_resets(MultiWalkerEnv, 3, seed=0x1234567890ABCDEF, gid0=1000, n=24, prefix="multiwalker_resetdraws_")
    record_philox_resets(MultiWalkerEnv, 2, seed=7, gid0=0, n=12, prefix="multiwalker_resetdraws_")
    record_philox_resets(MultiWalkerEnv, 10, seed=0xFEEDFACE, gid0=77, n=6, prefix="multiwalker_resetdraws_")
    record_philox_resets(MultiWalkerEnv, 6, seed=5, gid0=3, n=6, prefix="multiwalker_resetdraws_")
    return 0


if __name__ == "__main__":
    sys.exit(main())
