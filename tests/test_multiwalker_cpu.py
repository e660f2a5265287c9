"""CPU tests (-m "not gpu") of the MultiWalker dynamics.
PARITY UNPINNED: Box2D, where the reference's arithmetic for this env lives, is not available.  What IS checked:
  * the INDEPENDENT Box2D-2.3.0-ordered restatement (oracle/multiwalker_ref.c: plain C, Box2D's own lists / islands / sweeps, no
    code shared with the product) against the one published output of the library, the manual's "Hello Box2D" lines;
  * the PRODUCT's solver source (madrl_amd/csrc/multiwalker_core.hpp, compiled by g++: oracle/multiwalker_oracle.cpp) against
    that independent restatement, step by step: every body pose / velocity BIT FOR BIT, every contact of Box2D's world list
    (pair, order, touching, feature ids, warm-start impulses), every joint impulse, every ContactDetector flag, done, rewards;
  * physical invariants and the env logic around the solver."""
import os

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
        # (the independent Box2D-ordered restatement shows the same excursions bit for bit: test_product_source_matches_...)
        assert (knee > -1.6 - 1.3).all() and (knee < -0.1 + 1.3).all(), (knee.min(), knee.max())
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
            assert gap.max() < 0.45, gap.max()
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


def test_independent_oracle_replays_box2d_helloworld():
    """The one published numeric output of the absent dependency: Box2D v2.3 manual, "Hello Box2D" (a 2 x 2 box dropped from
    y = 4 onto static ground, 60 steps of 1/60 s, 6 / 2 iterations, printed "%4.2f %4.2f %4.2f"):
        0.00 4.00 0.00 / 0.00 3.99 0.00 / 0.00 3.98 0.00 / ... / 0.00 1.25 0.00 / 0.00 1.13 0.00 / 0.00 1.01 0.00
    replayed through the independent restatement's b2World::Step (polygon - polygon contact of a static and a dynamic body, island
    solve, continuous pass).  The box would reach y = 0.997 at step 46; the TOI sub-step leaves it at the published 1.01."""
    from oracle import multiwalker_ref as mwr
    for poly in (False, True):
        out = mwr.helloworld(60, poly=poly)
        line = lambda i: ("%4.2f %4.2f %4.2f" % tuple(out[i - 1])).replace("-0.00", "0.00")
        assert [line(1), line(2), line(3)] == ["0.00 4.00 0.00", "0.00 3.99 0.00", "0.00 3.98 0.00"]
        assert [line(44), line(45)] == ["0.00 1.25 0.00", "0.00 1.13 0.00"]
        assert line(46) == "0.00 1.01 0.00"                              # the sixth published line: needs the continuous pass
        assert all(line(i) == "0.00 1.01 0.00" for i in range(46, 61))   # ... and the box rests there
        # the resting height approaches polygonRadius * 2 - linearSlop = 0.015 above the surface from below, like Box2D's solver
        assert 1.0135 < out[59, 1] < 1.0151 and abs(out[59, 2]) < 1e-3


def _contacts_equal(a, b):
    (ai, af), (bi, bf) = a, b
    return len(ai) == len(bi) and np.array_equal(ai, bi) and np.array_equal(af, bf)


@pytest.mark.parametrize("n_walkers,reward_mech,descending", [(3, "local", False), (3, "local", True), (2, "global", False), (4, "local", True),
                                                              (4, "local", False), (1, "local", False),
                                                              # the capacity classes beyond four walkers (8 lanes per env: 5 .. 8, 16 lanes: 9, 10) --
                                                              # the reference's curriculum, lessons/multiwalker/env.yaml
                                                              (5, "local", False), (6, "global", True), (7, "local", True), (8, "local", False),
                                                              (9, "local", True), (10, "global", False), (10, "local", True)])
def test_product_source_matches_the_independent_oracle_bit_for_bit(n_walkers, reward_mech, descending):
    """The product's solver runs on four lanes per env (lane w: walker w's joints and contacts), scheduled so that constraints which share
    a body keep the island's order; the CPU build runs the lanes one after the other, in ascending or descending order -- the schedule
    must make that irrelevant.  Teacher-forced on the bodies (the product takes the oracle's poses and velocities at the start of every step; contacts,
    joint impulses, fat AABBs and sleep times are each side's own), random actions with stretches of zero actions (limp walkers
    collapse: hull contacts, game over, resets), auto-reset on done.  Both sides use the same sin / cos polynomial here."""
    from oracle import multiwalker_ref as mwr
    W, N, T = n_walkers, 24, 160
    ref = mwr.MultiWalkerRef(n_walkers=W, n_envs=N, seed=5, env_id_base=3, position_noise=0, angle_noise=0, reward_mech=reward_mech, poly=True)
    core = _mk(N, n_walkers=W, seed=5, env_id_base=3, reward_mech=reward_mech, lanes_descending=descending)
    ro, co = ref.reset(), core.reset()
    assert np.array_equal(ref.terrain(), core.terrain())
    # mass, inertia (the product stores the reciprocals: compare those)
    assert np.array_equal(np.float32(1) / ref.model()[:, 0], np.float32(1) / core.masses()[0::2]) and np.allclose(ref.model()[:, 1], core.masses()[1::2], rtol=1e-6)
    assert np.array_equal(ref.bodies(), core.bodies()[0]) and np.abs(ro - co).max() < 1e-6
    rng = np.random.RandomState(2)
    n_done = n_touch = 0
    for t in range(T):
        a = rng.uniform(-1, 1, (N, W, 4)).astype(np.float32)
        if t % 50 > 38:
            a[:] = 0
        core.set_bodies(ref.bodies())
        ro, rr, rd = ref.step(a)
        co, cr, cd = core.step(a)
        cb, cf = core.bodies()
        assert np.array_equal(ref.bodies(), cb), "step %d: body poses / velocities differ in their bits" % t
        assert np.array_equal(ref.flags(), cf) and np.array_equal(rd, cd), "step %d: ContactDetector flags / done" % t
        assert np.array_equal(ref.joints(), core.joints()), "step %d: joint impulses / limit states" % t
        assert np.array_equal(ref.aux(), core.aux()), "step %d: fat AABBs / sleep times / awake flags" % t
        for e in range(0, N, 5):
            assert _contacts_equal(ref.contacts(e), core.contacts(e)), "step %d env %d: the world's contact list" % (t, e)
            n_touch += int(ref.contacts(e)[0][:, 3].sum())
        assert np.abs(ro - co).max() <= 1e-6 * max(1.0, np.abs(ro).max()), "observations (float32 of the float64 expression)"
        assert np.abs(rr - cr).max() <= 1e-6 * max(1.0, np.abs(rr).max()), "rewards"
        n_done += int(rd.sum())
        if rd.any():
            ref.reset(mask=rd); core.reset(mask=rd)
    assert n_done > 0 and n_touch > 0 and ref.stats()["toi_events"] > 50
    assert not core.overflow().any(), "a contact did not fit the product's fixed-size contact storage (sticky Hot::overflow)"


def test_product_source_matches_the_independent_oracle_free_running():
    """No re-synchronisation at all: both are deterministic and identical in every bit, so 250 free-running steps stay identical."""
    from oracle import multiwalker_ref as mwr
    W, N = 3, 16
    ref = mwr.MultiWalkerRef(n_walkers=W, n_envs=N, seed=8, position_noise=0, angle_noise=0, poly=True)
    core = _mk(N, n_walkers=W, seed=8, lanes_descending=True)
    ref.reset(); core.reset()
    rng = np.random.RandomState(4)
    for t in range(250):
        a = rng.uniform(-1, 1, (N, W, 4)).astype(np.float32)
        ro, rr, rd = ref.step(a)
        co, cr, cd = core.step(a)
        assert np.array_equal(ref.bodies(), core.bodies()[0]) and np.array_equal(rd, cd), t
        if rd.any():
            ref.reset(mask=rd); core.reset(mask=rd)
    assert not core.overflow().any()


def test_injected_terrain_and_push_and_libm_sensitivity():
    """reset_with(terrain, push): the parity hooks a recorded Box2D episode would be replayed through.  And how sensitive this
    contact system is to the last bit of sin / cos: the independent oracle built with libm's sinf / cosf (what Box2D calls)
    instead of the product's polynomial stays within 1e-5 (relative to max(1, |x|)) of the product on most env-steps, and the
    rest are amplified rounding, not different contact sets -- the share is reported so that a future comparison with a real
    Box2D build is read against it."""
    from oracle import multiwalker_ref as mwr
    W, N, T = 3, 32, 120
    rng = np.random.RandomState(7)
    terrain = 400 / 30.0 / 4 + np.cumsum(rng.uniform(-0.03, 0.03, (N, 75)), axis=1) * (np.arange(75) > 20)
    push = rng.uniform(-5, 5, (N, W))
    ref = mwr.MultiWalkerRef(n_walkers=W, n_envs=N, seed=1, position_noise=0, angle_noise=0, poly=True)
    lib = mwr.MultiWalkerRef(n_walkers=W, n_envs=N, seed=1, position_noise=0, angle_noise=0, poly=False)
    core = _mk(N, n_walkers=W, seed=1)
    ref.reset(terrain=terrain, push=push); lib.reset(terrain=terrain, push=push); core.reset_with(terrain=terrain, push=push)
    assert np.array_equal(core.terrain(), terrain.astype(np.float32)) and np.array_equal(ref.bodies(), core.bodies()[0])
    within = []
    for t in range(T):
        a = rng.uniform(-1, 1, (N, W, 4)).astype(np.float32)
        core.set_bodies(lib.bodies()); ref.set_bodies(lib.bodies())
        lib.step(a); ref.step(a); core.step(a)
        assert np.array_equal(ref.bodies(), core.bodies()[0])
        lb = lib.bodies()
        within.append((np.abs(lb - core.bodies()[0]) / np.maximum(1.0, np.abs(lb))).reshape(N, -1).max(1) <= 1e-5)
    share = float(np.mean(within))
    print("libm-sin/cos oracle vs product: %.4f of env-steps within 1e-5" % share)
    assert share > 0.85


def test_sleeping_and_waking():
    """b2Island::Solve sleeping: limp walkers that have come to rest are put to sleep (velocities exactly zero) after half a
    second below the tolerances; apply_action wakes the walkers' bodies at the start of every step (SetMotorSpeed)."""
    o = _mk(4, seed=6, terminate_on_fall=False)
    o.reset()
    slept = 0
    for t in range(500):
        obs, rew, done = o.step(np.zeros((4, 3, 4)))
        aux = o.aux()
        asleep = aux[:, :, 5] == 0
        if asleep.any():
            slept += 1
            b = o.bodies()[0]
            assert (b[asleep][:, 3:] == 0).all(), "a sleeping body has zero velocity"
    assert slept > 0


@pytest.mark.parametrize("n_walkers", [3, 10])
def test_both_box2d_polygon_revisions_in_both_restatements(n_walkers):
    """b2CollidePolygons changed inside Box2D 2.3.x (2.3.0: hill-climbing b2FindMaxSeparation + 0.98 / 0.001 hysteresis; later: every edge normal +
    0.1 * b2_linearSlop), and the reference tree does not say which revision the authors' pybox2d wrapped.  Both restatements implement both
    (`polygon_revision`): under EITHER one the product source equals the independent oracle bit for bit, free-running; and the two revisions
    do differ -- hull / package contacts go through that routine every step -- on a share of env-steps that is worth knowing (a few per cent
    once walkers lie on the ground; teacher-forced count)."""
    from oracle import multiwalker_ref as mwr
    W, N, T = n_walkers, 32, 220
    kw = dict(n_walkers=W, n_envs=N, seed=17, position_noise=0, angle_noise=0, terminate_on_fall=False)
    refs = [mwr.MultiWalkerRef(poly=True, polygon_revision=r, **kw) for r in (0, 1)]
    core = [mwo.MultiWalkerOracle(position_noise=0.0, angle_noise=0.0, polygon_revision=r, lanes_descending=bool(r),
                                  **{k: v for k, v in kw.items() if k not in ("position_noise", "angle_noise")}) for r in (0, 1)]
    for o in refs + core:
        o.reset()
    probe = mwr.MultiWalkerRef(poly=True, polygon_revision=1, **kw)   # revision 1 teacher-forced on revision 0's states: how often ONE step differs
    probe.reset()
    rng = np.random.RandomState(5)
    differ = 0
    for t in range(T):
        a = rng.uniform(-1, 1, (N, W, 4)).astype(np.float32)
        if (t // 30) % 3 == 2:
            a[:] = 0
        probe.set_bodies(refs[0].bodies())
        for r in (0, 1):
            ro, rr, rd = refs[r].step(a)
            co, cr, cd = core[r].step(a)
            assert np.array_equal(refs[r].bodies(), core[r].bodies()[0]) and np.array_equal(rd, cd), (r, t)
            assert np.array_equal(refs[r].joints(), core[r].joints()) and np.array_equal(refs[r].aux(), core[r].aux()), (r, t)
            if rd.any():
                refs[r].reset(mask=rd); core[r].reset(mask=rd)
                if r == 0:
                    probe.reset(mask=rd)
        probe.step(a)
        differ += int((probe.bodies() != refs[0].bodies()).any(axis=(1, 2)).sum())
    assert not core[0].overflow().any() and not core[1].overflow().any()
    share = differ / float(N * T)
    print("n_walkers=%d: the two b2CollidePolygons revisions give different body states on %.2f %% of the env-steps" % (W, 100 * share))
    assert 0.0 < share < 0.25


@pytest.mark.parametrize("case", ["addpair_wakes", "update_reenables"])
def test_the_two_env_steps_a_longer_soak_caught(case):
    """tests/fixtures/mw_soak_finds.npz (oracle/make_fixture_mw_soak_finds.py): the product's world record of one env just before a step on
    which the product's source once disagreed with the independent restatement, and what the independent restatement has after that step.
    1. AddPair wakes both bodies (a hull the step's islands had just put to sleep); 2. b2Contact::Update re-enables a contact that an
    earlier event of the continuous pass had disabled (a leg tip on the vertex two terrain edges share).  The CPU build must land on the
    independent restatement's bits: bodies, joints, fat AABBs / sleep times / awake flags, contact flags."""
    import ctypes as C
    from oracle import multiwalker as mwo
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "fixtures", "mw_soak_finds.npz"))
    k = lambda name: g["%s_%s" % (case, name)]
    W = int(k("n_walkers"))
    for descending in (False, True):
        core = mwo.MultiWalkerOracle(n_walkers=W, n_envs=1, seed=int(k("seed")), position_noise=0.0, angle_noise=0.0, lanes_descending=descending)
        core.reset()
        w = np.ascontiguousarray(k("world")[None])
        assert w.shape == core.worlds().shape
        core.L.mwo_set_worlds(core.h, w.ctypes.data_as(C.c_void_p))
        core.step(k("actions")[None])
        bodies, flags = core.bodies()
        assert np.array_equal(bodies[0], k("bodies")), "body states"
        assert np.array_equal(core.joints()[0], k("joints")) and np.array_equal(core.aux()[0], k("aux")), "joints / fat AABBs, sleep times, awake flags"
        assert np.array_equal(np.asarray(flags[0], np.uint8), k("flags")) and not core.overflow().any()


@pytest.mark.parametrize("n_walkers", [3, 6, 10])
def test_an_episode_that_outlasts_its_contacts_creation_stamps_says_so(n_walkers):
    """A contact remembers the FindNewContacts call that created it in 16 bits (its place in Box2D's lists).  An episode with more calls than
    that -- ten fallen walkers with half a dozen continuous-pass events per step get there in 9 141 steps (scripts/mw_soak.py), walking
    ones in some 45 000 -- raises the record's sticky overflow flag (bit 3) 256 calls early instead of letting new contacts sort before
    old ones.  The call counter is poked into the raw record here."""
    import ctypes as C
    core = mwo.MultiWalkerOracle(n_walkers=n_walkers, n_envs=2, seed=5, position_noise=0.0, angle_noise=0.0)
    core.reset()
    off = (C.c_int32 * 2)()
    core.L.mwo_hot_offsets(off)
    w = core.worlds().copy()
    assert int(w[1, off[1]:off[1] + 4].view(np.uint32)[0]) < 100 and not w[:, off[0]].any()
    w[1, off[1]:off[1] + 4] = np.frombuffer(np.uint32(0xFEFE).tobytes(), np.uint8)
    core.L.mwo_set_worlds(core.h, w.ctypes.data_as(C.c_void_p))
    zero = np.zeros((2, n_walkers, 4), np.float32)
    core.step(zero)
    assert list(core.overflow()) == [0, 0]          # 0xFEFF calls: not yet
    core.step(zero)
    assert list(core.overflow() & 8) == [0, 8], "sticky bit 3 from 0xFF00 calls on"
    core.step(zero)
    assert list(core.overflow() & 8) == [0, 8]
    core.reset(mask=np.array([0, 1], np.uint8))
    assert not core.overflow().any()
