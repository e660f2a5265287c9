"""The MultiWalker ENV LAYER against the reference's own code.

tests/golden/multiwalker_envlayer_*.npz were recorded (oracle/make_golden_multiwalker.py --envlayer) from the UNMODIFIED
madrl_environments/walker/multi_walker.py running over oracle/shims_box2d: a package named `Box2D` whose b2World is the World of
oracle/multiwalker_ref.c.  So everything multi_walker.py itself does is the reference's code -- reset() building the package, the terrain
edges, hulls, legs and joints call by call (:113-192, :499-514, :613-620), the initial pushes (:130-131), apply_action (:194-203),
get_observation and LidarCallback (:183-190, :205-237), ContactDetector (:50-84), the neighbour / package part of the observation, the
shaping rewards, drop / fall rewards and the termination rules with the leaked `pos` of the last walker (:359-428) -- while b2World::Step
is the restated dynamics.  Each restatement of the env in this repository gets the same terrain, pushes and actions and runs FREE (no
teacher forcing): its world must stay identical to the recorded one in every bit of every body state -- which it only does if its reset
constructs the same world -- and its observations, rewards, ContactDetector flags and done must be the reference's.

What this does NOT show: that the dynamics are Box2D's.  PARITY of b2World::Step stays UNPINNED (no Box2D in the image).

  independent oracle (oracle/multiwalker_ref.c)          observations / rewards float64 like the reference: equal to 1e-12
  product source on the host (oracle/multiwalker_oracle.cpp), HIP kernels (-m gpu)     float32 observations / rewards: 1e-6 relative
"""
import glob
import os

import numpy as np
import pytest

from mw_scenes import Backend

FILES = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "multiwalker_envlayer_*.npz")))
ids = [os.path.basename(p)[len("multiwalker_envlayer_"):-4] for p in FILES]


def _replay(kind, path, obs_tol, rew_tol, copies=1):
    """copies > 1: every recorded episode `copies` times in one batch, interleaved (the kernels take 16 envs per wavefront: copies of one
    episode then sit in different lanes, wavefronts and blocks and must all come out the same).  Only without observation noise -- the
    noise is keyed by the env id, which a copy does not share."""
    g = np.load(path)
    W, E = int(g["n_walkers"]), int(g["n_episodes"]) * copies
    ep = [{k: g["ep%d_%s" % (e % (E // copies), k)] for k in ("terrain_y", "push", "actions", "bodies", "obs", "rew", "done", "flags")} for e in range(E)]
    assert copies == 1 or float(g["cfg_position_noise"]) == 0.0
    T = [len(r["done"]) for r in ep]
    be = Backend(kind, E, n_walkers=W, terminate_on_fall=bool(g["cfg_terminate_on_fall"]), reward_mech="global" if int(g["reward_global"]) else "local",
                 forward_reward=float(g["cfg_forward_reward"]), fall_reward=float(g["cfg_fall_reward"]), drop_reward=float(g["cfg_drop_reward"]),
                 one_hot=bool(g["cfg_one_hot"]), position_noise=float(g["cfg_position_noise"]), angle_noise=float(g["cfg_angle_noise"]),
                 seed=int(g["cfg_seed"]), env_id_base=int(g["cfg_env_id_base"]))
    obs = be.reset(terrain=np.stack([r["terrain_y"] for r in ep]), push=np.stack([r["push"] for r in ep]))
    close = lambda a, b, tol: np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max())
    for e in range(E):
        assert np.array_equal(be.bodies()[e].astype(np.float32), ep[e]["bodies"][0]), "episode %d: the world reset() built (+ its trailing step)" % e
        assert np.array_equal(np.asarray(be.flags()[e], np.uint8), ep[e]["flags"][0]), "episode %d: flags after reset" % e
        assert close(obs[e], ep[e]["obs"][0], obs_tol), "episode %d: reset observation (%g)" % (e, np.abs(obs[e] - ep[e]["obs"][0]).max())
    n_done = n_fallen = n_over = 0
    for t in range(max(T)):
        act = np.stack([r["actions"][t] if t < len(r["done"]) else np.zeros((W, 4), np.float32) for r in ep])
        obs, rew, done = be.step(act)
        bodies, flags = be.bodies(), be.flags()
        for e in range(E):
            if t >= T[e]:
                continue   # the recording of this episode ended (done); the env keeps stepping with zero actions, unobserved
            tag = "%s episode %d step %d" % (os.path.basename(path), e, t)
            assert np.array_equal(bodies[e].astype(np.float32), ep[e]["bodies"][t + 1]), tag + ": body states"
            assert np.array_equal(np.asarray(flags[e], np.uint8), ep[e]["flags"][t + 1]), tag + ": game_over / fallen / ground_contact"
            assert bool(done[e]) == bool(ep[e]["done"][t]), tag + ": done"
            assert close(obs[e], ep[e]["obs"][t + 1], obs_tol), tag + ": observations (%g)" % np.abs(obs[e] - ep[e]["obs"][t + 1]).max()
            assert close(rew[e], ep[e]["rew"][t], rew_tol), tag + ": rewards %r != %r" % (rew[e], ep[e]["rew"][t])
            n_done += int(ep[e]["done"][t]); n_fallen += int(ep[e]["flags"][t + 1][1:1 + W].any()); n_over += int(ep[e]["flags"][t + 1][0])
    return n_done, n_fallen, n_over


def test_recordings_cover_falls_drops_and_lidar_hits():
    assert len(FILES) >= 5
    seen = dict(done=0, fallen=0, over=0, lidar=0, ground=0, horizon=0)
    for path in FILES:
        g = np.load(path)
        W = int(g["n_walkers"])
        for e in range(int(g["n_episodes"])):
            fl, ob, dn = g["ep%d_flags" % e], g["ep%d_obs" % e], g["ep%d_done" % e]
            seen["done"] += int(dn.sum()); seen["fallen"] += int(fl[:, 1:1 + W].any()); seen["over"] += int(fl[:, 0].any())
            seen["lidar"] += int((ob[:, :, 14:24] < 1.0).sum()); seen["ground"] += int(fl[:, 1 + W:].sum()); seen["horizon"] += int(not dn.any())
    assert seen["done"] >= 10 and seen["fallen"] >= 8 and seen["over"] >= 2 and seen["lidar"] > 1000 and seen["ground"] > 500, seen


@pytest.mark.parametrize("path", FILES, ids=ids)
def test_independent_oracle_env_layer_is_the_references(path):
    _replay("ref", path, 1e-12, 1e-12)


@pytest.mark.parametrize("path", FILES, ids=ids)
def test_product_source_env_layer_is_the_references(path):
    _replay("core", path, 1e-6, 1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=ids)
def test_kernels_env_layer_is_the_references(path):
    _replay("hip", path, 1e-6, 1e-6)


def _noiseless(path):
    g = np.load(path)
    return float(g["cfg_position_noise"]) == 0.0 and float(g["cfg_angle_noise"]) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("path", [p for p in FILES if _noiseless(p)], ids=[i for p, i in zip(FILES, ids) if _noiseless(p)])
def test_kernels_env_layer_in_every_lane(path):
    """37 interleaved copies of every recorded episode: more than two wavefronts of envs per episode set, at every position of a wavefront"""
    _replay("hip", path, 1e-6, 1e-6, copies=37)


# ------------------------------------------------------------------------------------------------ the un-injected reset
DRAWS = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "multiwalker_resetdraws_*.npz")))


def _terrain_of(be):
    if be.kind == "hip":
        return be.o.get_state()["terrain"].cpu().numpy()
    return be.o.terrain()


def _reset_from_own_draws(kind, path):
    """multiwalker_resetdraws_*: the UNMODIFIED reference's reset() (its terrain walk with the grass counters, :516-612; package, walkers,
    pushes) fed, through a scripted generator, with the draws this repository's RNG contract assigns to (seed, env id, episode 0) -- so
    the plain reset() of every restatement, which makes those draws itself (Philox), must arrive at the same terrain, the same world
    and the same first observation."""
    g = np.load(path)
    W, n = int(g["n_walkers"]), len(g["terrain_y"])
    be = Backend(kind, n, n_walkers=W, seed=int(g["seed"]), env_id_base=int(g["env_id_base"]))
    obs = be.reset()
    assert np.array_equal(_terrain_of(be).astype(np.float32), g["terrain_y"].astype(np.float32)), "terrain heights"
    assert len(np.unique(g["terrain_y"][:, 30])) == n and np.abs(g["push"]).max() <= 5.0 and len(np.unique(g["push"])) == n * W
    assert np.array_equal(be.bodies().astype(np.float32), g["bodies"]), "world after reset (terrain, pushes, construction, trailing step)"
    assert np.array_equal(np.asarray(be.flags(), np.uint8), g["flags"])
    tol = 1e-12 if kind == "ref" else 1e-6
    assert np.abs(obs - g["obs"]).max() <= tol, np.abs(obs - g["obs"]).max()


@pytest.mark.parametrize("kind", ["ref", "core"])
@pytest.mark.parametrize("path", DRAWS, ids=[os.path.basename(p)[len("multiwalker_resetdraws_"):-4] for p in DRAWS])
def test_plain_reset_builds_what_the_reference_builds_from_the_same_draws(path, kind):
    _reset_from_own_draws(kind, path)


@pytest.mark.gpu
@pytest.mark.parametrize("path", DRAWS, ids=[os.path.basename(p)[len("multiwalker_resetdraws_"):-4] for p in DRAWS])
def test_kernel_reset_builds_what_the_reference_builds_from_the_same_draws(path):
    _reset_from_own_draws("hip", path)
