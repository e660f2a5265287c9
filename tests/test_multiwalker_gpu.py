"""GPU tests (-m gpu) for the HIP MultiWalkerEnv path through the C ABI.

PARITY UNPINNED against Box2D (see madrl_amd/csrc/multiwalker_core.hpp).  What is checked here:
(1) the GPU execution of the solver against the CPU build of the same source: whole world records, byte for byte;
(2) the HIP kernel against the INDEPENDENT Box2D-ordered restatement (oracle/multiwalker_ref.c) through the C ABI's
get_state / set_state / reset_with; (3) physical invariants at the BASELINE batch size; (4) the API."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-5


def _mk(n_envs, **kw):
    from madrl_amd.multiwalker import BatchedMultiWalkerEnv
    kw.setdefault("n_walkers", 3)
    kw.setdefault("position_noise", 0.0)
    kw.setdefault("angle_noise", 0.0)
    return BatchedMultiWalkerEnv(n_envs=n_envs, device=DEV, **kw)


@pytest.mark.parametrize("n_walkers,reward_mech", [(3, "local"), (2, "global"), (4, "local"), (1, "local"), (3, "one_hot"),
                                                   # the capacity classes beyond four walkers (lessons/multiwalker/env.yaml runs 2 .. 10):
                                                   # eight lanes per env for 5 .. 8 walkers, sixteen for 9 and 10
                                                   (5, "local"), (6, "global"), (7, "local"), (8, "local"), (9, "global"), (10, "local"), (10, "one_hot"),
                                                   # the whole step as ONE launch, every capacity class (since round 6 the sixteen-lane one too)
                                                   (3, "fused"), (8, "fused"), (9, "fused"), (10, "fused")])
def test_hip_matches_cpu_build_bit_for_bit(n_walkers, reward_mech):
    """The GPU execution (four lanes per env, sixteen envs per wavefront, the step in three launches, contact cache in HBM) against the
    CPU build of the same source (the lanes one after the other): the WHOLE per-env world record -- bodies, joints, every contact with its list position and impulses,
    fat AABBs, sleep times, flags -- must come out identical in every byte, every step.  (This checks the port; the algorithm is
    checked against the independent oracle below and, without a GPU, in tests/test_multiwalker_cpu.py.)"""
    from oracle import multiwalker as mwo
    N, T = 96, 70
    one_hot = reward_mech == "one_hot"   # ids as np.eye(MAX_AGENTS)[i] (multi_walker.py:397-398): 71-wide rows
    fused = reward_mech == "fused"
    reward_mech = "local" if one_hot or fused else reward_mech
    env = _mk(N, n_walkers=n_walkers, reward_mech=reward_mech, seed=11, env_id_base=7, one_hot=one_hot)
    if fused:
        env.set_mode(fused=True)
    orc = mwo.MultiWalkerOracle(n_walkers=n_walkers, position_noise=0.0, angle_noise=0.0, reward_mech=reward_mech,
                                n_envs=N, seed=11, env_id_base=7, one_hot=one_hot)
    if one_hot:
        assert env.obs_dim == 71 and env.agents[0].observation_space.shape == (71,)
    assert env.world_bytes >= orc.world_bytes   # the device record is followed by the step's scratch (manifolds, schedule)
    obs = env.reset()
    oobs = orc.reset()
    assert np.array_equal(obs.cpu().numpy(), oobs), "reset obs"
    assert np.array_equal(env.state_buffer.cpu().numpy()[:, :orc.world_bytes], orc.worlds()), "world records after reset"
    rng = np.random.RandomState(3)
    for t in range(T):
        act = rng.uniform(-1, 1, (N, n_walkers, 4)).astype(np.float32)
        if t % 30 > 22:
            act[:] = 0
        obs, rew, done, _ = env.step(act)
        oobs, orew, odone = orc.step(act)
        assert np.array_equal(done.cpu().numpy(), odone.astype(bool)), "step %d: done flags differ" % t
        assert np.array_equal(obs.cpu().numpy(), oobs) and np.array_equal(rew.cpu().numpy(), orew), "step %d: observations / rewards" % t
        same = (env.state_buffer.cpu().numpy()[:, :orc.world_bytes] == orc.worlds()).all(axis=1)
        assert same.all(), "step %d: %d world records differ" % (t, int((~same).sum()))
        if odone.any():
            orc.reset(mask=odone)
            env.reset(mask=odone)


@pytest.mark.parametrize("horizon,fused,n_walkers", [(0, False, 3), (9, False, 3), (0, True, 3), (7, False, 6), (0, True, 10), (11, False, 10)])
def test_auto_reset_through_spares_matches_mask_resets(horizon, fused, n_walkers):
    """auto_reset=True: an env whose episode ends gets its next episode from the spare record prepared ahead of time (multiwalker.hip), or,
    when the spare is not ready, from the second launch -- either way exactly what reset(mask) + the next steps give on the CPU build.  With
    a horizon every env ends its episode in the same call, and again `horizon` calls later: all spares consumed and rebuilt at once."""
    from oracle import multiwalker as mwo
    N, W, T = 80, n_walkers, 90
    env = _mk(N, n_walkers=W, seed=21, env_id_base=5, auto_reset=True, max_steps=horizon)
    if fused:
        env.set_mode(fused=True)   # the whole step in one launch: same results
    orc = mwo.MultiWalkerOracle(n_walkers=W, position_noise=0.0, angle_noise=0.0, n_envs=N, seed=21, env_id_base=5)
    assert np.array_equal(env.reset().cpu().numpy(), orc.reset())
    rng = np.random.RandomState(6)
    tstep = np.zeros(N, np.int64)
    n_resets = n_back_to_back = 0
    last = np.zeros(N, bool)
    for t in range(T):
        act = rng.uniform(-1, 1, (N, W, 4)).astype(np.float32)
        if t % 25 > 17:
            act[:] = 0
        obs, rew, done, info = env.step(act)
        oobs, orew, odone = orc.step(act)
        tstep += 1
        ends = odone.astype(bool) | ((tstep >= horizon) if horizon else False)
        assert np.array_equal(info["done_bits"].cpu().numpy() != 0, ends), "step %d: which episodes ended" % t
        assert np.array_equal(rew.cpu().numpy(), orew), "step %d: rewards" % t
        if ends.any():
            orc.reset(mask=ends.astype(np.uint8))
            tstep[ends] = 0
        n_resets += int(ends.sum()); n_back_to_back += int((ends & last).sum()); last = ends
        assert np.array_equal(obs.cpu().numpy(), orc.obs), "step %d: observations (the next episode's first one where an episode ended)" % t
        same = (env.state_buffer.cpu().numpy()[:, :orc.world_bytes] == orc.worlds()).all(axis=1)
        assert same.all(), "step %d: %d world records differ" % (t, int((~same).sum()))
    assert n_resets > N // 2 and (horizon == 0 or n_resets >= N * (T // horizon))


@pytest.mark.parametrize("n_walkers", [3, 2, 5, 8, 10])
def test_hip_matches_the_independent_oracle(n_walkers):
    """HIP kernel through the C ABI (get_state / set_state / reset_with) against oracle/multiwalker_ref.c -- the independent
    Box2D-2.3.0-ordered restatement -- teacher-forced on the bodies: tolerance 1e-5 as north_star asks, zero flag / done
    disagreements; in fact the body states agree in every bit (asserted) because both follow the same operation order."""
    from oracle import multiwalker_ref as mwr
    W, N, T = n_walkers, 64, 120
    rng = np.random.RandomState(9)
    terrain = 400 / 30.0 / 4 + np.cumsum(rng.uniform(-0.03, 0.03, (N, 25 * W)), axis=1) * (np.arange(25 * W) > 20)
    push = rng.uniform(-5, 5, (N, W))
    env = _mk(N, n_walkers=W, seed=2)
    ref = mwr.MultiWalkerRef(n_walkers=W, n_envs=N, seed=2, position_noise=0, angle_noise=0, poly=True)
    obs = env.reset_with(terrain=terrain, push=push)
    robs = ref.reset(terrain=terrain, push=push)
    assert np.abs(obs.cpu().numpy() - robs).max() <= TOL
    n_done = 0
    for t in range(T):
        a = rng.uniform(-1, 1, (N, W, 4)).astype(np.float32)
        if t % 40 > 30:
            a[:] = 0
        env.set_state(bodies=ref.bodies())
        obs, rew, done, _ = env.step(a)
        robs, rrew, rdone = ref.step(a)
        st = env.get_state()
        assert np.array_equal(done.cpu().numpy(), rdone.astype(bool)), "step %d: done" % t
        assert np.array_equal(st["flags"].cpu().numpy()[:, :1 + 3 * W], ref.flags()), "step %d: game_over / fallen / ground_contact" % t
        assert not st["flags"].cpu().numpy()[:, 1 + 3 * W].any(), "a contact did not fit its cache / the manifold pool"
        gb, rb = st["bodies"].cpu().numpy(), ref.bodies()
        assert np.abs(gb - rb).max() <= TOL and np.array_equal(gb, rb), "step %d: bodies, max |d| = %g" % (t, np.abs(gb - rb).max())
        assert np.array_equal(st["joints"].cpu().numpy(), ref.joints()) and np.array_equal(st["aux"].cpu().numpy(), ref.aux())
        assert np.abs(obs.cpu().numpy() - robs).max() <= TOL * max(1.0, np.abs(robs).max())
        assert np.abs(rew.cpu().numpy() - rrew).max() <= TOL * max(1.0, np.abs(rrew).max())
        n_done += int(rdone.sum())
        if rdone.any():
            ref.reset(mask=rdone, terrain=terrain, push=push)
            env.reset_with(mask=rdone, terrain=terrain, push=push)
    assert n_done > 0 and ref.stats()["toi_events"] > 100


def test_full_batch_invariants_c4():
    """BASELINE C4 (16 384 envs, n_walkers = 3)."""
    N = 16384
    env = _mk(N, seed=2, auto_reset=True, max_steps=100)
    obs = env.reset()
    assert obs.shape == (N, 3, 32) and torch.isfinite(obs).all()
    b0, f0, ty = env.bodies()
    assert torch.allclose(ty[:, :21], torch.full_like(ty[:, :21], 400 / 30.0 / 4), atol=1e-5)
    g = torch.Generator(device=DEV).manual_seed(0)
    n_done = 0
    for t in range(100):
        a = torch.rand((N, 3, 4), generator=g, device=DEV) * 2 - 1
        obs, rew, done, info = env.step(a)
        n_done += int(done.sum())
        assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
    b, f, _ = env.bodies()
    assert torch.isfinite(b).all()
    assert (b[..., 3:5].abs() < 40).all()                         # nothing explodes
    assert (b[:, :, 1] > 2.0).all() and (b[:, :, 1] < 9.0).all()  # everything stays near the terrain
    lid = obs[..., 14:24]
    assert (lid > 0).all() and (lid <= 1).all()
    hip, knee = obs[..., [4, 9]], obs[..., [6, 11]] - 1.0
    # joint limits hold up to the one-step overshoot Box2D's limit handling allows; the tail is
    # bounded loosely, the bulk tightly
    assert (hip.abs() < 3.2).all() and (knee.abs() < 3.2).all()
    inside = ((hip > -0.8 - 0.3) & (hip < 1.1 + 0.3)).float().mean(), ((knee > -1.6 - 0.3) & (knee < -0.1 + 0.3)).float().mean()
    # knees: a hard foot strike is resolved by the continuous (TOI) sub-step, which solves no joints (b2Island::SolveTOI)
    assert inside[0] > 0.999 and inside[1] > 0.98, inside
    assert n_done > 0                                              # random flailing makes some walkers fall


def test_determinism_and_launch_shape():
    outs = []
    for blocks in (0, 37):   # (max_blocks: accepted and ignored since the launches became one wavefront per group of envs)
        env = _mk(300, seed=4, max_blocks=blocks)
        env.reset()
        g = torch.Generator(device="cpu").manual_seed(1)
        for t in range(15):
            a = (torch.rand((300, 3, 4), generator=g) * 2 - 1).to(DEV)
            obs, rew, done, _ = env.step(a)
        outs.append((obs.cpu().clone(), env.state_buffer[:, :env.world_bytes].cpu().clone()))   # the world records (the step scratch behind them holds
        # the manifolds in pool order, i.e. in the order the lanes' atomic allocations happened to land)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_n1_dropin_api_matches_reference_types():
    from madrl_amd.multiwalker import MultiWalkerEnv
    env = MultiWalkerEnv(n_walkers=3, reward_mech="local", device=DEV)   # multi_walker.py:644-648
    assert len(env.agents) == 3 and env.agents[0].observation_space.shape == (32,) and env.agents[0].action_space.shape == (4,)
    obs = env.reset()
    assert isinstance(obs, list) and len(obs) == 3 and obs[0].shape == (32,) and obs[0].dtype == np.float64
    a = np.array([env.agents[0].action_space.sample() for _ in range(3)])
    o, r, done, info = env.step(a)
    assert isinstance(r, np.ndarray) and r.shape == (3,) and isinstance(done, bool) and info == {}
    genv = MultiWalkerEnv(n_walkers=2, reward_mech="global", device=DEV)
    o, r, done, info = genv.step(np.zeros(8))
    assert isinstance(r, list) and len(r) == 2 and r[0] == r[1]


def test_the_reference_curriculum_walks_through_the_capacity_classes():
    """lessons/multiwalker/env.yaml: n_walkers 2, 3, ..., 10, applied by runners/curriculum.py:56-91 through set_param_values -> setup()
    (madrl_environments/__init__.py:64-67).  Here a new walker count may mean another capacity class of the kernels: another handle over
    another state buffer.  Every lesson must then behave like a freshly constructed env of that size (CPU build, byte for byte), and the N == 1
    drop-in must follow too."""
    from madrl_amd.multiwalker import MultiWalkerEnv
    from oracle import multiwalker as mwo
    N = 24
    env = _mk(N, n_walkers=2, seed=13)
    classes = []
    rng = np.random.RandomState(8)
    for W in (2, 3, 4, 5, 6, 7, 8, 9, 10, 3):
        env.set_param_values({"n_walkers": W})
        assert env.n_bodies == 5 * W + 1 and env.n_terrain == 200 * W // 8 and len(env.agents) == W
        assert abs(env.package_length - 240 / 30.0 * W / 1.75) < 1e-12           # multi_walker.py:293-294
        classes.append((env.capacity_walkers, env.lanes_per_env))
        orc = mwo.MultiWalkerOracle(n_walkers=W, position_noise=0.0, angle_noise=0.0, n_envs=N, seed=13)
        assert np.array_equal(env.reset().cpu().numpy(), orc.reset())
        for t in range(6):
            a = rng.uniform(-1, 1, (N, W, 4)).astype(np.float32)
            obs, rew, done, _ = env.step(a)
            oobs, orew, odone = orc.step(a)
            assert np.array_equal(obs.cpu().numpy(), oobs) and np.array_equal(rew.cpu().numpy(), orew) and np.array_equal(done.cpu().numpy(), odone.astype(bool))
        assert (env.state_buffer.cpu().numpy()[:, :orc.world_bytes] == orc.worlds()).all()
    assert classes == [(4, 4)] * 3 + [(8, 8)] * 4 + [(10, 16)] * 2 + [(4, 4)]
    one = MultiWalkerEnv(n_walkers=2, device=DEV)
    one.set_param_values({"n_walkers": 9})
    obs = one.reset()
    assert len(obs) == 9 and len(one.agents) == 9
    o, r, d, info = one.step(np.zeros((9, 4)))
    assert len(o) == 9 and len(r) == 9


@pytest.mark.parametrize("n_walkers", [3, 9])
def test_later_box2d_polygon_revision_on_the_kernels(n_walkers):
    """box2d_polygon_revision=1 (b2CollidePolygons as later Box2D 2.3.x revisions have it; include/madrl_hip.h madrl_multiwalker_config): the kernels
    equal the CPU build of the same source byte for byte, as they do for the default"""
    from oracle import multiwalker as mwo
    N, W, T = 64, n_walkers, 80
    env = _mk(N, n_walkers=W, seed=31, terminate_on_fall=False, box2d_polygon_revision=1)
    orc = mwo.MultiWalkerOracle(n_walkers=W, position_noise=0.0, angle_noise=0.0, n_envs=N, seed=31, terminate_on_fall=False, polygon_revision=1)
    assert np.array_equal(env.reset().cpu().numpy(), orc.reset())
    rng = np.random.RandomState(3)
    for t in range(T):
        a = rng.uniform(-1, 1, (N, W, 4)).astype(np.float32)
        if t % 30 > 18:
            a[:] = 0
        obs, rew, done, _ = env.step(a)
        oobs, orew, odone = orc.step(a)
        assert np.array_equal(obs.cpu().numpy(), oobs) and np.array_equal(rew.cpu().numpy(), orew) and np.array_equal(done.cpu().numpy(), odone.astype(bool)), t
        assert (env.state_buffer.cpu().numpy()[:, :orc.world_bytes] == orc.worlds()).all(), t
        if odone.any():
            orc.reset(mask=odone); env.reset(mask=odone)


@pytest.mark.parametrize("n_walkers,capacity", [(3, 4), (6, 8), (10, 10)])
def test_capacity_overflow_is_reported_in_the_done_byte(n_walkers, capacity):
    """A contact that does not fit its cache or the step's manifold pool is ignored and the env's sticky Hot::overflow flag set (in gait
    rollouts with terminate_on_fall off and eight walkers lying in a heap: twice in 192 000 env-steps, scripts/mw_soak.py).  The env must
    SAY so in-band: bit 7 of the done byte / info['overflow'] -- the flag get_state returns -- every step until the env is reset, no other
    bit set by it, no new episode started.  (The flag is poked into the raw record here; its offset is checked against the CPU build.)"""
    import ctypes as C
    from oracle import multiwalker as mwo
    N, W = 64, n_walkers
    off = 24 * (5 * capacity + 1) + 4 * capacity
    off = (off + 7) // 8 * 8 + 8 * capacity + 8 + 3 * capacity + 1     # Hot: bodies, push_x, prev_shaping[], prev_package_shaping, fallen, ground, game_over, overflow
    orc = mwo.MultiWalkerOracle(n_walkers=W, n_envs=2, seed=1, position_noise=0.0, angle_noise=0.0)
    orc.reset()
    w = orc.worlds().copy()
    assert not w[:, off].any()
    w[1, off] = 1
    orc.L.mwo_set_worlds(orc.h, w.ctypes.data_as(C.c_void_p))
    assert list(orc.overflow() != 0) == [False, True], "the offset of Hot::overflow"
    env = _mk(N, n_walkers=W, seed=3, auto_reset=True, max_steps=0)
    env.reset()
    zero = torch.zeros((N, W, 4), device=DEV)
    obs, rew, done, info = env.step(zero)
    assert not info["overflow"].any()
    marked = torch.zeros(N, dtype=torch.bool, device=DEV)
    marked[[5, 40]] = True
    env.state_buffer[marked, off] = 1
    for _ in range(3):
        obs, rew, done, info = env.step(zero)
        assert torch.equal(info["overflow"], marked) and torch.equal(env.get_state()["flags"][:, -1] != 0, marked)
        assert not ((info["done_bits"] & 0x7C) != 0).any() and torch.equal((info["done_bits"] & 1).bool(), done)
        assert not done[marked].any()      # standing walkers, zero actions: nothing ends, and the flag ends nothing
    env.reset(mask=marked.to(torch.uint8))
    obs, rew, done, info = env.step(zero)
    assert not info["overflow"].any(), "reset clears the flag"
