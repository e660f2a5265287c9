"""CPU tests (-m "not gpu") of the MultiWalker dynamics (CPU build of the solver source).
PARITY UNPINNED: Box2D, where the reference's arithmetic for this env lives, is not available
(SURVEY.md 8(c)); these tests check physical invariants and the env logic around the solver."""
import numpy as np
import pytest

from oracle import multiwalker as mwo

SCALE = 30.0
LEG_H = 34 / SCALE
TERRAIN_STEP = 14 / SCALE


def _mk(n_envs=4, **kw):
    kw.setdefault("n_walkers", 3)
    kw.setdefault("position_noise", 0.0)
    kw.setdefault("angle_noise", 0.0)
    return mwo.MultiWalkerOracle(n_envs=n_envs, **kw)


def test_mass_properties_match_box2d_formulas():
    m = _mk(1).masses()
    # hull: polygon area of HULL_POLY / SCALE^2 times density 5; legs: boxes with density 1
    hull = np.array([(-30, 9), (6, 9), (34, 1), (34, -8), (-30, -8)], float) / SCALE
    x, y = hull[:, 0], hull[:, 1]
    area = 0.5 * abs(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1)))
    assert abs(m[2] - 5.0 * area) < 1e-4
    assert abs(m[4] - (8 / SCALE) * LEG_H) < 1e-5 and abs(m[6] - 0.8 * (8 / SCALE) * LEG_H) < 1e-5
    w, h = 8 / SCALE, LEG_H
    assert abs(m[5] - m[4] * (w * w + h * h) / 12) < 1e-5   # box inertia about its centre


def test_reset_layout_and_terrain():
    o = _mk(8, seed=3)
    obs = o.reset()
    assert obs.shape == (8, 3, 32) and np.isfinite(obs).all()
    ty = o.terrain()
    assert ty.shape == (8, 75)                                     # int(200 * 3 / 8)
    assert np.allclose(ty[:, :21], 400 / SCALE / 4, atol=1e-5)     # flat start pad (:532)
    assert (np.abs(np.diff(ty, axis=1)) < 0.25).all() and ty[:, 21:].std() > 0.01
    b, f = o.bodies()
    # walkers start WALKER_SEPERATION terrain steps apart, package centred above them
    hx = b[:, [1, 6, 11], 0]
    assert np.allclose(np.diff(hx, axis=1), 10 * TERRAIN_STEP, atol=0.05)
    assert np.allclose(b[:, 0, 0], hx.mean(1), atol=0.05)
    assert (obs[..., 31] == np.array([0, 1, 2], np.float32) / 3).all()   # id = i / n_walkers (:400)
    # lidar fractions are in (0, 1]; the first ray points straight down at the ground
    lid = obs[..., 14:24]
    assert (lid > 0).all() and (lid <= 1).all() and (lid[..., 0] < 0.6).all()


def test_free_fall_matches_gravity_before_contact():
    """A body in free flight must follow v_y(t) = v_y(0) - 10 t exactly (semi-implicit Euler)."""
    o = _mk(1, seed=1)
    o.reset()
    b0, _ = o.bodies()
    o.step(np.zeros((1, 3, 4)))
    b1, _ = o.bodies()
    # the package starts 3 * LEG_H above the terrain and the hulls are below it: it is falling freely
    assert abs((b1[0, 0, 4] - b0[0, 0, 4]) - (-10.0 / 50)) < 1e-5
    assert abs((b1[0, 0, 1] - b0[0, 0, 1]) - b1[0, 0, 4] / 50) < 1e-5


def test_constraints_hold_under_random_actions():
    """Joint anchors stay together, joint angles stay inside their limits (plus slop), nothing
    sinks through the terrain, nothing explodes."""
    o = _mk(8, seed=5)
    o.reset()
    rng = np.random.RandomState(0)
    max_v = 0.0
    knee_over, anchor_gaps = [], []
    for t in range(150):
        obs, rew, done = o.step(rng.uniform(-1, 1, (8, 3, 4)))
        b, f = o.bodies()
        assert np.isfinite(b).all() and np.isfinite(obs).all() and np.isfinite(rew).all()
        max_v = max(max_v, np.abs(b[..., 3:5]).max())
        # joint angles: obs[4] = hip angle in [-0.8, 1.1], obs[6] - 1 = knee angle in [-1.6, -0.1]
        hip = obs[..., [4, 9]]
        knee = obs[..., [6, 11]] - 1.0
        # limits are enforced from the step AFTER the crossing (b2RevoluteJoint sets its limit state at the start of a step): one step of overshoot is legal
        assert (hip > -0.8 - 0.4).all() and (hip < 1.1 + 0.4).all(), (hip.min(), hip.max())
        # knees: a lower leg that hits the terrain is moved by the continuous (TOI) sub-step, whose island holds contacts only
        # -- Box2D solves no joints there (b2Island::SolveTOI) -- so a hard foot strike may leave the knee beyond its limit
        # until the next step's joint position correction: bounded absolutely, rare beyond 0.4 rad
        assert (knee > -1.6 - 1.5).all() and (knee < -0.1 + 1.5).all(), (knee.min(), knee.max())
        knee_over.append(np.maximum(knee - (-0.1), -1.6 - knee).clip(0))
        # hip anchor: hull origin + R(hull) (0, LEG_DOWN)  ==  upper-leg centre + R(leg) (0, LEG_H / 2)
        for w in range(3):
            hull, up = b[:, 1 + 5 * w], b[:, 2 + 5 * w]
            # hull centre of mass is offset from its origin; compare through the second anchor instead:
            lo = b[:, 3 + 5 * w]
            a_up = up[:, :2] + np.stack([np.sin(up[:, 2]) * (LEG_H / 2), -np.cos(up[:, 2]) * (LEG_H / 2)], 1)   # bottom of upper leg
            a_lo = lo[:, :2] + np.stack([-np.sin(lo[:, 2]) * (LEG_H / 2), np.cos(lo[:, 2]) * (LEG_H / 2)], 1)  # top of lower leg
            # "Continuous collision does not handle joints ... you may see joint stretching on fast moving objects" (Box2D manual):
            # the TOI sub-step of a foot strike moves the lower leg alone; the joint is pulled together again over the next steps
            gap = np.abs(a_up - a_lo).max(axis=1)
            assert gap.max() < 0.6, gap.max()
            anchor_gaps.append(gap)
        if done.any():
            o.reset(mask=done)
    assert max_v < 30.0
    assert (np.stack(knee_over) > 0.4).mean() < 0.02
    assert (np.concatenate(anchor_gaps) > 0.03).mean() < 0.15 and np.median(np.concatenate(anchor_gaps)) < 0.01
    ty = o.terrain()
    b, f = o.bodies()
    # lower legs do not sink below the terrain by more than a few slops
    for w in range(3):
        for k in (3, 5):
            lo = b[:, k + 5 * w]
            foot = lo[:, :2] + np.stack([np.sin(lo[:, 2]) * (LEG_H / 2), -np.cos(lo[:, 2]) * (LEG_H / 2)], 1)
            idx = np.clip((foot[:, 0] / TERRAIN_STEP).astype(int), 0, 73)
            ground = ty[np.arange(8), idx]
            assert (foot[:, 1] > ground - 0.25).all()


def test_termination_flags_and_rewards():
    o = _mk(4, seed=2)
    o.reset()
    fell = dropped = False
    for t in range(400):
        obs, rew, done = o.step(np.zeros((4, 3, 4)))   # limp walkers collapse
        b, f = o.bodies()
        if f[:, 1:4].any():
            fell = True
            n = np.nonzero(f[:, 1:4].any(1))[0][0]
            assert done[n] == 1 and (rew[n][f[n, 1:4] == 1] < -90).all()   # fall_reward, terminate_on_fall
        if f[:, 0].any():
            dropped = True
            n = np.nonzero(f[:, 0])[0][0]
            assert done[n] == 1 and (rew[n] < -90).all()                    # drop_reward for everybody
        if done.all():
            break
    assert fell


def test_global_reward_is_the_mean_and_determinism():
    a = _mk(4, seed=9, reward_mech="local")
    g = _mk(4, seed=9, reward_mech="global")
    a.reset(); g.reset()
    rng = np.random.RandomState(1)
    for t in range(20):
        act = rng.uniform(-1, 1, (4, 3, 4))
        _, ra, _ = a.step(act)
        _, rg, _ = g.step(act)
        assert np.allclose(rg, ra.mean(1, keepdims=True).repeat(3, 1), atol=1e-5)
    assert np.array_equal(a.worlds(), _replay(9, 20))


def _replay(seed, steps):
    o = _mk(4, seed=seed, reward_mech="local")
    o.reset()
    rng = np.random.RandomState(1)
    for t in range(steps):
        o.step(rng.uniform(-1, 1, (4, 3, 4)))
    return o.worlds()


def test_observation_noise_has_the_requested_scale():
    q = _mk(256, seed=4, position_noise=0.0, angle_noise=0.0)
    n = _mk(256, seed=4, position_noise=1e-2, angle_noise=1e-2)
    oq = q.reset(); on = n.reset()
    d = (on - oq)[:, 1, 24:31]            # middle walker: 2 neighbours (4 values) + package (3 values)
    assert np.abs(d.mean(0)).max() < 3e-3
    assert np.allclose(d.std(0), 1e-2, rtol=0.2)
    assert np.array_equal(on[..., :24], oq[..., :24])


def test_box2d_helloworld_known_answer():
    """The one published numeric output of the absent dependency: Box2D v2.3 manual, "Hello Box2D" (a 2x2 box dropped from
    y = 4 onto static ground, 60 steps of 1/60 s, 6/2 iterations, printed "%4.2f %4.2f %4.2f"):
        0.00 4.00 0.00 / 0.00 3.99 0.00 / 0.00 3.98 0.00 / ... / 0.00 1.25 0.00 / 0.00 1.13 0.00 / 0.00 1.01 0.00
    replayed through the env's own world_step (oracle/box2d_kat.cpp).  Steps 44-46 are the three tail lines: the box
    would reach y = 0.997 at step 46; the continuous (TOI) pass stops it at the surface and its sub-step leaves it at the
    published 1.01.  All six published lines are asserted."""
    import ctypes
    import os
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_build", "libmadrl_b2kat.so")
    if not os.path.exists(so):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.dirname(os.path.dirname(so))])
    L = ctypes.CDLL(so)
    out = np.zeros((60, 3), np.float32)
    assert L.b2kat_falling_box(out.ctypes.data_as(ctypes.c_void_p), 60) == 0
    line = lambda i: ("%4.2f %4.2f %4.2f" % tuple(out[i - 1])).replace("-0.00", "0.00")
    assert [line(1), line(2), line(3)] == ["0.00 4.00 0.00", "0.00 3.99 0.00", "0.00 3.98 0.00"]
    assert [line(44), line(45)] == ["0.00 1.25 0.00", "0.00 1.13 0.00"]
    assert line(46) == "0.00 1.01 0.00"                              # the sixth published line: needs the continuous pass
    assert all(line(i) == "0.00 1.01 0.00" for i in range(46, 61))   # ... and the box rests there
    # the resting height approaches polygonRadius * 2 - linearSlop = 0.015 above the surface from below, like Box2D's solver
    assert 1.0135 < out[59, 1] < 1.015 and abs(out[59, 2]) < 1e-3
