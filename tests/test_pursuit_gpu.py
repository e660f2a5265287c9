"""GPU parity tests (-m gpu): the HIP PursuitEvade path, called through the C ABI, against
(1) the golden vectors of the unmodified reference, (2) the C oracle on seeded free-running
rollouts (independent Philox implementations), (3) size-independent properties at the
BASELINE batch sizes.  Bit-exact everywhere: positions, flags, counts, observations; rewards
are float32 roundings of the reference's float64 values."""
import numpy as np
import pytest
import torch

from helpers import pursuit_golden_files, golden_id

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _mk(maps, n_envs, **kw):
    from madrl_amd.pursuit import BatchedPursuitEvade
    return BatchedPursuitEvade(maps, n_envs=n_envs, device=DEV, **kw)


def _cmp_state(st_gpu, st_ref, msg=""):
    for k in ("pos_p", "pos_e", "gone", "term_p", "term_e", "map_id"):
        a = st_gpu[k].cpu().numpy()
        b = st_ref[k]
        assert np.array_equal(a, b), "%s: state[%s] differs in %d entries" % (msg, k, int((a != b).sum()))
    assert np.array_equal(st_gpu["tick"].cpu().numpy().view(np.uint32), st_ref["tick"]), msg + ": tick"


WAVE_GOLDEN = {"pursuit_c1_surround_local", "pursuit_c1_surround_global", "pursuit_c1_colocate_hwc",
               "pursuit_pool16_sample_maps", "pursuit_tiny5_dense", "pursuit_in_building",
               "pursuit_nonsquare_12x20", "pursuit_window_gt_map", "pursuit_random_opponents",
               "pursuit_c5_32x32",  # two wavefronts per env (pursuit_group.hpp)
               "pursuit_authors_30v50_obs11", "pursuit_authors_30v30_obs11",   # ... with 22 float4 slots per thread (the LDS slot table, round 6)
               "pursuit_fuzz_13", "pursuit_fuzz_20"}  # the drawn configurations whose shape the one-wavefront kernel can take (odd obs_range, <= 8 float4 slots per lane)


@pytest.mark.parametrize("path", pursuit_golden_files(), ids=golden_id)
@pytest.mark.parametrize("kernel", ["generic64", "generic128", "wave"])
def test_hip_matches_reference_golden(path, kernel):
    """Replay the reference's own recorded episodes (injected positions / evader actions)
    through both kernel implementations."""
    from oracle import pursuit as po
    g = np.load(path)
    N = 3  # identical copies: also catches cross-env indexing mistakes
    if kernel == "wave":
        if golden_id(path) not in WAVE_GOLDEN:
            pytest.skip("no one-wavefront specialisation for this shape (generic kernel covers it)")
        env = _mk(list(g["maps"]), N, kernel="wave", **po.config_from_golden(g))
        assert env.kernel_kind == "wave"
    else:
        env = _mk(list(g["maps"]), N, kernel="generic", threads=0 if kernel == "generic64" else 128,
                  **po.config_from_golden(g))
        assert env.kernel_kind == "generic"
    rep = lambda a: np.repeat(np.asarray(a)[None], N, axis=0)
    for t in range(len(g["op"])):
        want_obs = g["obs_f32"][t].reshape(env.n_pursuers, -1)
        if g["op"][t] == 0:
            pos = np.concatenate([g["init_p"][t], g["init_e"][t]])
            obs = env.reset(positions=rep(pos), map_ids=np.full(N, g["map_id"][t]))
            got = obs.reshape(N, env.n_pursuers, -1).cpu().numpy()
            for n in range(N):
                assert np.array_equal(got[n], want_obs), "%s reset obs op %d env %d" % (golden_id(path), t, n)
        else:
            obs, rew, done, info = env.step(rep(g["act_p"][t]), evader_actions=rep(g["act_e"][t]))
            got = obs.reshape(N, env.n_pursuers, -1).cpu().numpy()
            st = env.get_state()
            for n in range(N):
                tag = "%s op %d env %d" % (golden_id(path), t, n)
                assert np.array_equal(got[n], want_obs), tag + ": obs (%d cells differ)" % int((got[n] != want_obs).sum())
                assert np.array_equal(rew[n].cpu().numpy(), g["rew_f64"][t].astype(np.float32)), tag + ": rewards"
                assert bool(done[n]) == bool(g["done"][t]), tag + ": done"
                assert int(info["removed"][n]) == int(g["removed"][t]), tag + ": removed"
                assert np.array_equal(st["pos_p"][n].cpu().numpy(), g["pos_p"][t]), tag + ": pursuer positions"
                assert np.array_equal(st["pos_e"][n].cpu().numpy(), g["pos_e"][t]), tag + ": evader positions"
                assert np.array_equal(st["gone"][n].cpu().numpy(), g["gone_e"][t]), tag + ": evaders_gone"


CASES = {
    "c2_surround": dict(maps="rect16", n_pursuers=8, n_evaders=30, obs_range=7, n_catch=2, surround=True, flatten=True, reward_mech="local"),
    "c2_global_coloc": dict(maps="rect16", n_pursuers=8, n_evaders=30, obs_range=7, n_catch=1, surround=False, flatten=False,
                            reward_mech="global", catchr=0.1, urgency_reward=-0.1),
    "pool16": dict(maps="pool16", n_pursuers=8, n_evaders=30, obs_range=7, n_catch=2, surround=True, flatten=True,
                   reward_mech="local", sample_maps=True),
    "c5_32x32": dict(maps="rect32", n_pursuers=16, n_evaders=60, obs_range=7, n_catch=2, surround=True, flatten=True, reward_mech="local"),
    "c2_random_opponents": dict(maps="rect16", n_pursuers=8, n_evaders=30, obs_range=7, n_catch=2, surround=True, flatten=True,
                                reward_mech="local", random_opponents=True, max_opponents=25),
    "group_20v50_pool": dict(maps="pool16", n_pursuers=20, n_evaders=50, obs_range=5, n_catch=1, surround=False, flatten=True,
                             reward_mech="global", sample_maps=True, random_opponents=True, max_opponents=45, catchr=0.1),
    "group_20v50_surround": dict(maps="pool16", n_pursuers=20, n_evaders=50, obs_range=5, n_catch=2, surround=True, flatten=True,
                                 reward_mech="local", sample_maps=True),
    # the authors' largest launch line: runners/old/rllab/pursuit_cnn.sh:1 (100 pursuers / 300 evaders, obs_range 21, 128 x 128; their
    # map_pool128.npy is not in the tree -> rectangle_map(128, 128))
    "authors_cnn_100v300": dict(maps="rect128", n_pursuers=100, n_evaders=300, obs_range=21, n_catch=2, surround=True, flatten=False,
                                reward_mech="local", n_envs=6, steps=20, expect_catches=False),   # (random pursuers surround nobody on a 128 x 128 map in 20 steps)
    # the authors' own training shapes (runners/old/rllab/pursuit.sh:1, runners/old/rltools/pursuit.sh:1) on resize(2, map_pool16): the
    # two-wavefront kernel with the LDS slot table (22 slots per thread, three stale-zero mask words)
    "authors_30v50_obs11": dict(maps="pool32", n_pursuers=30, n_evaders=50, obs_range=11, n_catch=2, surround=True, flatten=True,
                                reward_mech="local", sample_maps=True, n_envs=192, steps=90),
    "authors_30v30_obs11": dict(maps="pool32", n_pursuers=30, n_evaders=30, obs_range=11, n_catch=2, surround=True, flatten=True,
                                reward_mech="local", sample_maps=True, catchr=0.0, term_pursuit=5.0, n_envs=192, steps=60),
    "tiny_window": dict(maps="open6", n_pursuers=5, n_evaders=4, obs_range=5, n_catch=2, surround=True, flatten=True,
                        reward_mech="global", constraint_window=0.5),
}


def _maps(name):
    from madrl_amd.maps import rectangle_map
    if name == "rect16":
        return [rectangle_map(16, 16)]
    if name == "rect32":
        return [rectangle_map(32, 32)]
    if name == "rect128":
        return [rectangle_map(128, 128)]
    if name == "open6":
        return [np.zeros((6, 6), np.int32)]
    if name == "pool32":   # TwoDMaps.resize(2, map_pool16), as recorded in the authors' shape goldens
        g = np.load(pursuit_golden_files()[[golden_id(p) for p in pursuit_golden_files()].index("pursuit_authors_30v50_obs11")])
        return list(g["maps"])
    if name == "pool16":
        g = np.load(pursuit_golden_files()[[golden_id(p) for p in pursuit_golden_files()].index("pursuit_pool16_sample_maps")])
        return list(g["maps"])
    raise KeyError(name)


@pytest.mark.parametrize("case", sorted(CASES), ids=sorted(CASES))
@pytest.mark.parametrize("kernel", ["generic", "auto"])
def test_hip_matches_oracle_free_running(case, kernel):
    """Seeded free-running rollouts with auto-reset: HIP kernels and the C oracle each run their
    own Philox; every output and the whole state must agree on every step."""
    kw = dict(CASES[case])
    maps = _maps(kw.pop("maps"))
    N, T = kw.pop("n_envs", 512), kw.pop("steps", 120)
    expect_catches = kw.pop("expect_catches", True)
    if kernel == "auto" and kw["n_pursuers"] + kw["n_evaders"] > 128:
        pytest.skip("no fast path above two wavefronts of agents: the generic kernel is what runs")
    _free_run(maps, kw, kernel, N, T, expect_catches, want_kind="wave" if kernel == "auto" else None)


def _drawn_cases(n=16, seed=20260927):
    """configurations drawn like the recorded ones of oracle/make_golden_pursuit_fuzz.py, from another seed: here nothing is injected --
    reset sampling, map draws, random_opponents and the evaders' moves all come from the Philox contract on both sides"""
    from oracle.make_golden_pursuit_fuzz import draw_case
    rng = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        maps, cfg, _run = draw_case(rng)
        if rng.rand() < 0.3:
            cfg["constraint_window"] = float(rng.choice([0.3, 0.5, 0.8]))
        out.append((maps, cfg))
    return out


@pytest.mark.parametrize("i", range(16))
def test_drawn_configurations_free_running_vs_oracle(i):
    maps, cfg = _drawn_cases()[i]
    big = cfg["n_pursuers"] + cfg["n_evaders"] > 64
    _free_run(maps, cfg, "auto", 96 if big else 256, 50, expect_catches=False)


def _free_run(maps, kw, kernel, N, T, expect_catches, want_kind=None, H=25):
    from oracle import pursuit as po
    env = _mk(maps, N, seed=2024, env_id_base=1000, max_steps=H, auto_reset=True, kernel=kernel, **kw)
    if want_kind:
        assert env.kernel_kind == want_kind  # one wavefront per env, or a wavefront group for more than 64 agents
    orc = po.PursuitOracle(maps, n_envs=N, seed=2024, env_id_base=1000, **kw)
    obs = env.reset()
    oobs = orc.reset().copy()
    assert np.array_equal(obs.reshape(oobs.shape).cpu().numpy(), oobs), "reset obs"
    _cmp_state(env.get_state(), orc.get_state(), "after reset")
    rng = np.random.RandomState(5)
    tstep = np.zeros(N, np.int64)
    n_done = n_removed = 0
    for t in range(T):
        act = rng.randint(5, size=(N, env.n_pursuers))
        obs, rew, done, info = env.step(torch.as_tensor(act, device=DEV))
        oobs, orew, odone, orem = orc.step(act)
        tstep += 1
        bits = odone.astype(np.uint8) | ((tstep >= H).astype(np.uint8) << 1)
        assert np.array_equal(info["done_bits"].cpu().numpy(), bits), "step %d done bits" % t   # (bit 7, count overflow, never set)
        assert np.array_equal(info["removed"].cpu().numpy(), orem), "step %d removed" % t
        assert np.array_equal(rew.cpu().numpy(), orew.astype(np.float32)), "step %d rewards" % t
        mask = (bits != 0).astype(np.uint8)
        if mask.any():
            orc.reset(mask=mask)  # rewrites the masked rows of orc.obs (shared persistent buffer)
            tstep[mask != 0] = 0
        n_done += int((bits & 1).sum())
        n_removed += int(orem.sum())
        got = obs.reshape(orc.obs.shape).cpu().numpy()
        assert np.array_equal(got, orc.obs), "step %d obs: %d cells differ" % (t, int((got != orc.obs).sum()))
        if t % 10 == 0 or t == T - 1:
            _cmp_state(env.get_state(), orc.get_state(), "step %d" % t)
            assert np.array_equal(env.get_state()["t"].cpu().numpy(), tstep), "episode step counter"
    assert n_removed > 0 or not expect_catches


@pytest.mark.parametrize("shape", ["c2_wave", "c5_group"])
def test_bit31_of_the_alive_and_terminal_masks(shape):
    """Evader slot 31 gone / agent 31 terminal: bit 31 of a record word must not leak into the upper half of the
    64-bit wave masks (v_readlane returns a signed int).  Fast path against the generic kernel from the same state."""
    from madrl_amd.maps import rectangle_map
    if shape == "c2_wave":
        maps, P, E = [rectangle_map(16, 16)], 8, 30
    else:
        maps, P, E = [rectangle_map(32, 32)], 16, 60
    N = 64
    kw = dict(n_pursuers=P, n_evaders=E, obs_range=7, n_catch=2, surround=True, flatten=True, seed=4)
    envs = [_mk(maps, N, kernel=k, **kw) for k in ("generic", "wave")]
    for env in envs:
        env.reset()
        st = env.get_state()
        gone = torch.zeros((N, E), dtype=torch.uint8)
        term_e = torch.zeros((N, E), dtype=torch.uint8)
        if E > 31:
            gone[:, 31] = 1
        term_e[:, 31 - P] = 1            # agent index 31
        if P + E > 64:
            term_e[:, 95 - P if 95 - P < E else E - 1] = 1
            gone[:, E - 1] = 1
        env.set_state(dict(gone=gone, term_e=term_e))
    g = torch.Generator(device="cpu").manual_seed(3)
    for t in range(12):
        act = torch.randint(0, 5, (N, P), generator=g, dtype=torch.int32).to(DEV)
        outs = [env.step(act) for env in envs]
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), "step %d" % t
        sa, sb = envs[0].get_state(), envs[1].get_state()
        for k in sa:
            assert torch.equal(sa[k], sb[k]), "step %d state[%s]" % (t, k)
    assert bool(sa["term_e"][:, 31 - P].all()) and (E <= 31 or bool(sa["gone"][:, 31].all()))


def test_launch_shape_does_not_change_results():
    from madrl_amd.maps import rectangle_map
    kw = dict(n_pursuers=8, n_evaders=30, obs_range=7, surround=True, reward_mech="local", seed=9, max_steps=40, auto_reset=True)
    outs = []
    for kernel, threads, blocks in (("generic", 64, 0), ("generic", 128, 0), ("generic", 256, 7), ("generic", 64, 300),
                                    ("wave", 0, 0), ("wave", 0, 13)):
        env = _mk([rectangle_map(16, 16)], 1000, kernel=kernel, threads=threads, max_blocks=blocks, **kw)
        env.reset()
        g = torch.Generator(device="cpu").manual_seed(1)
        for _ in range(60):
            act = torch.randint(0, 5, (1000, 8), generator=g, dtype=torch.int32)
            obs, rew, done, info = env.step(act.to(DEV))
        st = env.get_state()
        outs.append((obs.cpu().clone(), rew.cpu().clone(), {k: v.cpu() for k, v in st.items()}))
    for o in outs[1:]:
        assert torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1])
        for k in o[2]:
            assert torch.equal(o[2][k], outs[0][2][k]), k


def test_walk_direction_does_not_change_results():
    """large batches alternate the direction in which a launch walks the envs (memory-side cache reuse); results are the same"""
    outs = []
    for walk in ("forward", "alternate"):
        from madrl_amd.maps import rectangle_map
        env = _mk([rectangle_map(16, 16)], 4099, seed=21, max_steps=30, auto_reset=True, n_pursuers=8, n_evaders=30, obs_range=7,
                  n_catch=2, surround=True, flatten=True)
        env.set_walk(walk)
        assert env.kernel_kind == "wave"
        env.reset()
        g = torch.Generator(device="cpu").manual_seed(0)
        acc = []
        for t in range(45):
            a = torch.randint(0, 5, (4099, 8), generator=g, dtype=torch.int32).to(DEV)
            o, r, d, info = env.step(a)
            acc.append((o.clone(), r.clone(), d.clone(), info["removed"].clone()))
        outs.append(acc)
    for (o1, r1, d1, m1), (o2, r2, d2, m2) in zip(*outs):
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2) and torch.equal(m1, m2)


def test_sharding_is_invisible_env_id_base():
    """Env n of a shard with env_id_base=b behaves exactly like env b+n of one big batch
    (this is what makes multi-GPU sharding need no communication)."""
    from madrl_amd.maps import rectangle_map
    kw = dict(n_pursuers=8, n_evaders=30, obs_range=7, surround=True, reward_mech="local", seed=3, max_steps=30, auto_reset=True)
    full = _mk([rectangle_map(16, 16)], 512, **kw)
    half = _mk([rectangle_map(16, 16)], 256, env_id_base=256, **kw)
    full.reset(); half.reset()
    g = torch.Generator(device="cpu").manual_seed(4)
    for _ in range(50):
        act = torch.randint(0, 5, (512, 8), generator=g, dtype=torch.int32).to(DEV)
        o1, r1, d1, _ = full.step(act)
        o2, r2, d2, _ = half.step(act[256:].contiguous())
    assert torch.equal(o1[256:], o2) and torch.equal(r1[256:], r2) and torch.equal(d1[256:], d2)


def test_full_batch_invariants_c2():
    """BASELINE C2 size (65 536 envs): size-independent properties of a free-running rollout."""
    from madrl_amd.maps import rectangle_map
    m = rectangle_map(16, 16)
    N, P, E = 65536, 8, 30
    env = _mk([m], N, n_pursuers=P, n_evaders=E, obs_range=7, surround=True, reward_mech="local", seed=11)
    mt = torch.as_tensor(m, device=DEV)
    obs = env.reset()
    st = env.get_state()
    assert (mt[st["pos_p"][..., 0].long(), st["pos_p"][..., 1].long()] == 0).all()
    assert (mt[st["pos_e"][..., 0].long(), st["pos_e"][..., 1].long()] == 0).all()
    # reset distribution: uniform over free cells (chi-square over 2M evader draws)
    cnt = torch.zeros(256, device=DEV, dtype=torch.float64)
    cell = (st["pos_e"][..., 0].long() * 16 + st["pos_e"][..., 1].long()).flatten()
    cnt.scatter_add_(0, cell, torch.ones_like(cell, dtype=torch.float64))
    free = (mt == 0).flatten()
    exp = cnt.sum() / free.sum()
    chi2 = float((((cnt[free] - exp) ** 2) / exp).sum())
    dof = int(free.sum()) - 1
    assert abs(chi2 - dof) < 6 * (2 * dof) ** 0.5, (chi2, dof)
    assert float(cnt[~free].sum()) == 0
    gone_prev = st["gone"].clone()
    total_removed = torch.zeros(N, dtype=torch.int64, device=DEV)
    g = torch.Generator(device=DEV).manual_seed(0)
    for t in range(40):
        act = torch.randint(0, 5, (N, P), generator=g, device=DEV, dtype=torch.int32)
        obs, rew, done, info = env.step(act)
        total_removed += info["removed"].long()
    st = env.get_state()
    alive = st["gone"] == 0
    assert (st["gone"] >= gone_prev).all()                                  # evaders never come back
    assert torch.equal(total_removed, st["gone"].long().sum(1))             # info['removed'] adds up
    x, y = st["pos_e"][..., 0], st["pos_e"][..., 1]
    assert ((x >= 0) & (x < 16) & (y >= 0) & (y < 16))[alive].all()
    assert (mt[x.clamp(min=0).long(), y.clamp(min=0).long()] == 0)[alive].all()   # nobody inside a building
    assert ((x == -1) & (y == -1))[~alive].all()
    assert (mt[st["pos_p"][..., 0].long(), st["pos_p"][..., 1].long()] == 0).all()
    # observation row structure: id column, channel-0 centre is the pursuer's own (free) cell,
    # channel 1 centre counts at least the pursuer itself
    o = obs.view(N, P, 148)
    ids = torch.arange(P, device=DEV, dtype=torch.float64) / P
    assert torch.equal(o[..., 147], ids.float().expand(N, P))
    c = o[..., :147].view(N, P, 3, 7, 7)
    assert (c[:, :, 0, 3, 3] == 0).all()
    assert (c[:, :, 1, 3, 3] >= 0.1).all()
    assert total_removed.sum() > 0


def test_n1_dropin_api_matches_reference_types():
    from madrl_amd.pursuit import PursuitEvade
    from madrl_amd.maps import rectangle_map
    env = PursuitEvade([rectangle_map(16, 16)], n_evaders=30, n_pursuers=8, obs_range=7, n_catch=2, surround=True,
                       flatten=True, reward_mech="local")
    assert len(env.agents) == 8 and env.agents[0].observation_space.shape == (148,) and env.agents[0].action_space.n == 5
    obs = env.reset()
    assert isinstance(obs, list) and len(obs) == 8 and obs[0].shape == (148,) and obs[0].dtype == np.float64
    obs, rew, done, info = env.step([0, 1, 2, 3, 4, 0, 1, 2])
    assert isinstance(obs, list) and isinstance(rew, np.ndarray) and rew.shape == (8,)
    assert isinstance(done, bool) and set(info) == {"removed"} and isinstance(info["removed"], int)
    obs, rew, done, info = env.step(np.array([4] * 8))
    obs, rew, done, info = env.step(1234)  # joint scalar action, pursuit_evade.py:231-235
    genv = PursuitEvade([rectangle_map(16, 16)], n_evaders=30, n_pursuers=8, obs_range=7, reward_mech="global")
    genv.reset()
    _, rew, _, _ = genv.step([4] * 8)
    assert isinstance(rew, list) and len(rew) == 8 and len(set(rew)) == 1
    with pytest.raises(IndexError):
        env.step([7] * 8)


def test_pickle_roundtrip_and_param_updates():
    import pickle
    from madrl_amd.maps import rectangle_map
    env = _mk([rectangle_map(16, 16)], 16, n_pursuers=8, n_evaders=30, obs_range=7, reward_mech="local", seed=5)
    env.set_param_values(dict(catchr=0.5, constraint_window=0.5))
    env2 = pickle.loads(pickle.dumps(env))
    assert env2.catchr == 0.5 and env2.constraint_window == 0.5 and env2.n_envs == 16
    o1 = env.reset(); o2 = env2.reset()
    assert torch.equal(o1, o2)


from helpers import evadercontrol_golden_files, replay_evadercontrol  # noqa: E402


@pytest.mark.parametrize("kernel", ["generic", "wave"])
@pytest.mark.parametrize("path", evadercontrol_golden_files(), ids=golden_id)
def test_hip_matches_reference_golden_evader_control(path, kernel):
    """train_pursuit=False through the C ABI against the record of the unmodified reference: the generic kernel and the
    one-wavefront-per-env kernel's evader-control instantiation."""
    from oracle import pursuit as po
    g = np.load(path)
    env = _mk(list(g["maps"]), 1, kernel=kernel, **po.config_from_golden(g))
    assert env.kernel_kind == kernel and not env.train_pursuit

    def state():
        st = env.get_state()
        return dict(pos_p=st["pos_p"][0].cpu().numpy(), pos_e=st["pos_e"][0].cpu().numpy(), gone=st["gone"][0].cpu().numpy())

    def step(aa, ao):
        obs, rew, done, info = env.step(torch.as_tensor(aa, device=DEV), evader_actions=torch.as_tensor(ao, device=DEV))
        return obs[0].cpu().numpy(), rew[0].cpu().numpy(), int(info["done_bits"][0]) & 1, int(info["removed"][0])

    replay_evadercontrol(g, lambda pos: env.reset(positions=pos)[0].cpu().numpy(), step, state)


@pytest.mark.parametrize("kernel,size", [("generic", 12), ("wave", 16), ("generic", 16)])
def test_evader_control_free_running_and_dropin_types(kernel, size):
    """train_pursuit=False free-running (in-kernel Philox for the pursuers) against the C oracle, 256 envs with auto-reset; then
    the N == 1 drop-in's return types: None entries for the gone evader slots below n_pursuers (pursuit_evade.py:418-428)."""
    from oracle import pursuit as po
    from madrl_amd.maps import rectangle_map
    from madrl_amd.pursuit import PursuitEvade
    maps = [rectangle_map(size, size)]     # 16 x 16, 5 v 9, obs_range 5 has a one-wavefront specialisation
    kw = dict(n_pursuers=5, n_evaders=9, obs_range=5, n_catch=2, surround=True, flatten=True, reward_mech="local", train_pursuit=False)
    N, T, H = 256, 80, 20
    env = _mk(maps, N, seed=6, env_id_base=40, max_steps=H, auto_reset=True, kernel=kernel, **kw)
    assert env.kernel_kind == kernel
    orc = po.PursuitOracle(maps, n_envs=N, seed=6, env_id_base=40, **kw)
    assert np.array_equal(env.reset().cpu().numpy(), orc.reset())
    rng = np.random.RandomState(2)
    tstep = np.zeros(N, np.int64)
    removed = 0
    for t in range(T):
        act = rng.randint(5, size=(N, 5))
        obs, rew, done, info = env.step(torch.as_tensor(act, device=DEV))
        oobs, orew, odone, orem = orc.step(act)
        tstep += 1
        bits = odone.astype(np.uint8) | ((tstep >= H).astype(np.uint8) << 1)
        assert np.array_equal(info["done_bits"].cpu().numpy(), bits) and np.array_equal(info["removed"].cpu().numpy(), orem)
        assert np.array_equal(rew.cpu().numpy(), orew.astype(np.float32))
        mask = (bits != 0).astype(np.uint8)
        if mask.any():
            orc.reset(mask=mask)
            tstep[mask != 0] = 0
        removed += int(orem.sum())
        assert np.array_equal(obs.cpu().numpy(), orc.obs), "step %d" % t   # rows past the last observer stay stale on both sides
        valid = env.obs_rows_valid().cpu().numpy()
        assert np.array_equal(valid.sum(1), 5 - orc.get_state()["gone"][:, :5].sum(1))
    _cmp_state(env.get_state(), orc.get_state(), "end")
    assert removed > 0 or size == 16   # (five random pursuers surround nobody on the larger map; catches on this path: the golden replays)
    one = PursuitEvade(maps, **kw)
    assert len(one.agents) == 5
    obs = one.reset()
    assert len(obs) == 5 and all(o.shape == (76,) for o in obs)
    one._env.set_state(dict(gone=np.array([[0, 1, 0, 0, 1, 0, 0, 0, 0]], np.uint8)))
    obs, rew, done, info = one.step([4] * 5)
    assert [o is None for o in obs] == [False, True, False, False, True] and isinstance(rew, np.ndarray) and rew.shape == (5,)
    assert obs[2][75] == np.float32(1 / 5.0)   # id = index in the compacted layer (1: slot 1 is gone) / n_pursuers


@pytest.mark.parametrize("kernel", ["wave", "generic"])
def test_in_place_edits_of_the_returned_observations_are_noticed(kernel):
    """step() / reset() return the persistent IN / OUT buffer itself -- the reference's local_obs (pursuit_evade.py:119-120), which it hands
    out as views in its (R, R, 4) mode and as copies in its flatten mode (:441-449).  A caller that edits the tensor in place -- a
    normaliser's `obs.sub_(mean)` -- changes the never-stored cells outside the map for good, as an edit of the reference's views would;
    what it must NOT do is leave the fast path's stale-zero masks describing a buffer that no longer exists (they would write +0 over the
    edited cells).  PyTorch's per-tensor version counter tells the env (BatchedPursuitEvade._check_obs_untouched): the run below equals
    the oracle's, whose local_obs receives the edited buffer at every edit, bit for bit -- without any invalidate_obs() call."""
    from madrl_amd.maps import rectangle_map
    from madrl_amd.pursuit import BatchedPursuitEvade
    from oracle import pursuit as po
    maps = [rectangle_map(16, 16)]
    kw = dict(n_pursuers=8, n_evaders=30, obs_range=7, n_catch=2, surround=True, flatten=True, reward_mech="local")
    N, P, R = 192, 8, 7
    env = BatchedPursuitEvade(maps, n_envs=N, device=DEV, seed=5, **kw)
    env.set_kernel(kernel)
    orc = po.PursuitOracle(maps, n_envs=N, seed=5, **kw)
    obs, oobs = env.reset(), orc.reset()
    assert np.array_equal(obs.cpu().numpy(), oobs)
    rng = np.random.RandomState(4)
    edits = 0

    def mirror():   # the oracle's persistent buffer takes the values the edited tensor holds (channels 0 - 2 of every row; the id is not part of local_obs)
        lo = orc.local_obs()
        lo[:, :, :3] = obs.cpu().numpy()[:, :, :3 * R * R].reshape(N, P, 3, R, R).astype(np.float64)
        orc.set_local_obs(lo)

    for t in range(40):
        if t % 7 == 3:      # an in-place edit of everything: never-stored cells that held 0 hold 0.25 from here on
            obs.add_(0.25); mirror(); edits += 1
        if t % 7 == 5:
            obs.mul_(2.0); mirror(); edits += 1
        if t == 20:         # ... and through a view
            obs[:, 0].zero_(); mirror(); edits += 1
        act = rng.randint(5, size=(N, 8))
        obs, rew, done, info = env.step(torch.as_tensor(act, device=DEV))
        oobs, orew, odone, orem = orc.step(act)
        assert np.array_equal(obs.cpu().numpy(), oobs), "step %d: observations (never-stored cells included)" % t
        assert np.array_equal(rew.cpu().numpy(), orew.astype(np.float32))
        if odone.any():
            m = odone.astype(np.uint8)
            obs, oobs = env.reset(mask=m), orc.reset(mask=m)
            assert np.array_equal(obs.cpu().numpy(), oobs)
    g = obs.cpu().numpy()[:, :, R * R:3 * R * R]
    assert edits >= 10 and (np.abs(g) > 1.5).any(), "edited values survive in the never-stored cells (nothing in the map's range is that large)"
