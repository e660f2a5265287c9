"""GPU tests (-m gpu) for the round-5 review: step() results that are views of what the launch wrote (no torch kernel after the
launch), sub-batches stepped with join=False / fork=False, the overflow bit of a done byte is no episode boundary, an env built
under torch.inference_mode(), a policy with `out=` that also returns values."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
KW = dict(n_pursuers=8, n_evaders=30, obs_range=7, n_catch=2, surround=True, flatten=True)


def _env(n, **kw):
    from madrl_amd.maps import rectangle_map
    from madrl_amd.pursuit import BatchedPursuitEvade
    return BatchedPursuitEvade([rectangle_map(16, 16)], n_envs=n, device=DEV, **dict(KW, **kw))


@pytest.mark.parametrize("kernel", ["wave", "generic"])
def test_step_flags_are_views_of_the_launch_output(kernel):
    """done / truncated / count_overflow of BatchedPursuitEvade.step are bool views of the flag plane the step kernel writes
    (include/madrl_hip.h, madrl_pursuit_flags_offset): equal to the bits of the done byte, on both kernels, with fused auto-resets"""
    N = 2048
    env = _env(N, seed=3, max_steps=7, auto_reset=True, kernel=kernel)
    env.reset()
    g = torch.Generator(device="cpu").manual_seed(0)
    seen_done = seen_trunc = 0
    for t in range(30):
        a = torch.randint(0, 5, (N, 8), generator=g, dtype=torch.int32).to(DEV)
        obs, rew, done, info = env.step(a)
        bits = info["done_bits"]
        assert done.dtype == torch.bool and info["truncated"].dtype == torch.bool and info["count_overflow"].dtype == torch.bool
        assert torch.equal(done, (bits & 1) != 0) and torch.equal(info["truncated"], (bits & 2) != 0)
        assert torch.equal(info["count_overflow"], (bits & 0x80) != 0)
        assert done.data_ptr() == env._flags.data_ptr()          # a view of the state buffer's flag plane, not a fresh tensor
        assert torch.equal(env._flags[:, 3], bits)
        seen_done += int(done.sum())
        seen_trunc += int(info["truncated"].sum())
    assert seen_trunc > 0   # the horizon was crossed


def test_flags_of_the_group_kernel():
    from madrl_amd.maps import rectangle_map
    from madrl_amd.pursuit import BatchedPursuitEvade
    N = 512
    env = BatchedPursuitEvade([rectangle_map(32, 32)], n_envs=N, device=DEV, seed=1, max_steps=5, auto_reset=True, n_pursuers=16, n_evaders=60, obs_range=7)
    assert env.kernel_kind == "wave"
    env.reset()
    for t in range(11):
        a = torch.randint(0, 5, (N, 16), dtype=torch.int32, device=DEV)
        _, _, done, info = env.step(a)
        assert torch.equal(done, (info["done_bits"] & 1) != 0) and torch.equal(info["truncated"], (info["done_bits"] & 2) != 0)
        assert bool(info["truncated"].all()) == ((t + 1) % 5 == 0)


@pytest.mark.parametrize("fork,join,dtype", [(True, False, torch.int32), (False, False, torch.int32), (True, False, torch.int64), (False, True, torch.int64)])
def test_sub_batches_without_join_give_the_one_batch_results(fork, join, dtype):
    """StreamSharded.step(join=False) / (fork=False): nothing of the step may run on the caller's stream unordered against the
    sub-batch launches -- the returned done / truncated tensors and actions that need a dtype conversion included"""
    from madrl_amd.sharded import StreamSharded
    N, S = 4096, 2
    kw = dict(seed=9, max_steps=6, auto_reset=True)
    one = _env(N, env_id_base=50, **kw)
    sh = StreamSharded(lambda n_envs, env_id_base, device: _env(n_envs, env_id_base=env_id_base, **kw), N, n_streams=S, env_id_base=50, device=DEV)
    one.reset()
    sh.reset()
    g = torch.Generator(device="cpu").manual_seed(4)
    for t in range(20):
        a = torch.randint(0, 5, (N, 8), generator=g, dtype=torch.int32).to(DEV).to(dtype)
        if not fork:
            torch.cuda.synchronize()   # fork=False: the caller promises the actions are ready
        o1, r1, d1, i1 = one.step(a)
        parts = sh.step(a, fork=fork, join=join)
        if not join:
            sh.join()
        assert torch.equal(d1, torch.cat([p[2] for p in parts])), t
        assert torch.equal(i1["truncated"], torch.cat([p[3]["truncated"] for p in parts])), t
        assert torch.equal(i1["done_bits"], torch.cat([p[3]["done_bits"] for p in parts])), t
        assert torch.equal(r1, torch.cat([p[1] for p in parts])) and torch.equal(o1, torch.cat([p[0] for p in parts])), t


def test_overflow_bit_of_a_done_byte_is_no_episode_boundary():
    """bit 7 of a done byte is the sticky capacity report of the Pursuit / MultiWalker kernels: the return scan, the diagnostics
    accumulators and the observation buffer must not treat it as an episode end"""
    from madrl_amd import _lib
    from oracle.rollout_oracle import gae as gae_oracle
    T, N, A = 12, 64, 3
    rng = np.random.RandomState(0)
    rew = rng.randn(T, N, A).astype(np.float32)
    val = rng.randn(T + 1, N, A).astype(np.float32)
    done = (rng.rand(T, N) < 0.15).astype(np.uint8)
    noisy = done | np.where(rng.rand(T, N) < 0.5, 0x80, 0).astype(np.uint8)   # overflow reports sprinkled over it
    out = {}
    for name, d in (("clean", done), ("noisy", noisy)):
        r, dn, v = (torch.as_tensor(x, device=DEV) for x in (rew, d, val))
        ret, adv = torch.empty_like(r), torch.empty_like(r)
        _lib.check(_lib.lib().madrl_rollout_gae(_lib.ptr(r), _lib.ptr(dn), _lib.ptr(v), T, N, A, 0.97, 0.9, _lib.ptr(ret), _lib.ptr(adv),
                                                _lib.current_stream(torch.device(DEV))))
        out[name] = (ret.cpu().numpy(), adv.cpu().numpy())
    assert np.array_equal(out["clean"][0], out["noisy"][0]) and np.array_equal(out["clean"][1], out["noisy"][1])
    oret, oadv = gae_oracle(rew, done, val, 0.97, 0.9)
    assert np.allclose(out["noisy"][0], oret, atol=1e-5) and np.allclose(out["noisy"][1], oadv, atol=1e-5)
    # diagnostics: an env that only ever reports 0x80 never finishes an episode before max_traj_len
    f64 = lambda *s: torch.zeros(s, dtype=torch.float64, device=DEV)
    ep_r, ep_len, dret, dpow = f64(N, A), torch.zeros(N, dtype=torch.int32, device=DEV), f64(N), f64(N)
    o_r, o_d, o_len, o_fin = f64(N, A), f64(N), torch.zeros(N, dtype=torch.int32, device=DEV), torch.zeros(N, dtype=torch.uint8, device=DEV)
    r = torch.ones((N, A), dtype=torch.float32, device=DEV)
    dn = torch.full((N,), 0x80, dtype=torch.uint8, device=DEV)
    dn[::2] = 0x81
    for _ in range(3):
        _lib.check(_lib.lib().madrl_wrap_diagnostics(_lib.ptr(r), _lib.ptr(dn), _lib.ptr(ep_r), _lib.ptr(ep_len), _lib.ptr(dret), _lib.ptr(dpow), N, A, 0.99, 1000,
                                                     _lib.ptr(o_r), _lib.ptr(o_d), _lib.ptr(o_len), _lib.ptr(o_fin), _lib.current_stream(torch.device(DEV))))
    assert o_fin[::2].all() and not o_fin[1::2].any() and int(ep_len[1]) == 3 and int(ep_len[0]) == 0
    # observation buffer: 0x80 pushes (history shifts), 0x81 refills
    k, per = 4, 5
    obs = torch.arange(N * per, dtype=torch.float32, device=DEV)
    buf = torch.zeros((N * per, k), dtype=torch.float32, device=DEV)
    _lib.check(_lib.lib().madrl_wrap_obsbuffer(_lib.ptr(obs), _lib.ptr(buf), N * per, per, k, _lib.ptr(dn), None, _lib.current_stream(torch.device(DEV))))
    b = buf.view(N, per, k).cpu().numpy()
    o = obs.view(N, per).cpu().numpy()
    assert np.array_equal(b[0], np.repeat(o[0][:, None], k, 1))                       # 0x81: reset, all slots filled
    assert np.array_equal(b[1][:, :3], np.zeros((per, 3))) and np.array_equal(b[1][:, 3], o[1])   # 0x80: an ordinary push


def test_env_built_under_inference_mode_steps():
    """a tensor allocated under torch.inference_mode() keeps no version counter; the env falls back to "nothing known" about its
    observation buffer before every launch instead of raising"""
    N = 256
    with torch.inference_mode():
        env = _env(N, seed=2)
        o0 = env.reset().clone()
        a = torch.randint(0, 5, (N, 8), dtype=torch.int32, device=DEV)
        o1 = env.step(a)[0].clone()
    ref = _env(N, seed=2)
    assert torch.equal(ref.reset(), o0) and torch.equal(ref.step(a.clone())[0], o1)


def test_policy_with_out_and_values_keeps_its_values():
    """RolloutCollector: a policy that takes `out=` AND returns (actions, values) -- the values are stored at every step, GAE
    bootstraps over initialised values"""
    from madrl_amd.rollout import RolloutCollector
    N, T = 256, 9

    def plain(obs):
        s = obs.sum(dim=2)
        return (s.abs() * 5.1).to(torch.int32) % 5, torch.tanh(s * 0.02)

    def with_out(obs, out=None):
        a, v = plain(obs)
        if out is not None:
            out.copy_(a)
            return out, v
        return a, v

    ta = RolloutCollector(_env(N, seed=6, max_steps=5, auto_reset=True), plain, T, gae_lambda=0.9).collect()
    tb = RolloutCollector(_env(N, seed=6, max_steps=5, auto_reset=True), with_out, T, gae_lambda=0.9).collect()
    for k in ("actions", "rewards", "dones", "values", "returns", "advantages"):
        assert torch.equal(getattr(ta, k), getattr(tb, k)), k


def test_waterworld_and_hostage_done_is_a_view():
    from madrl_amd.waterworld import BatchedMAWaterWorld
    from madrl_amd.hostage import BatchedContinuousHostageWorld
    w = BatchedMAWaterWorld(5, 10, n_envs=128, device=DEV, seed=0, max_steps=3, auto_reset=True)
    w.reset()
    for t in range(4):
        _, _, done, info = w.step(torch.zeros((128, 5, 2), device=DEV))
        # (reset() ends with a step(zeros) that counts, waterworld.py:172: with a limit of 3 every second step() ends an episode)
        assert done.dtype == torch.bool and done.data_ptr() == w._done.data_ptr() and torch.equal(done, w._done != 0)
        assert bool(done.all()) == bool(done.any()) == (t % 2 == 1)
        assert info["done_bits"] is w._done
    h = BatchedContinuousHostageWorld(3, 10, 5, 2, 2, n_envs=128, device=DEV, seed=0, max_steps=3, auto_reset=True)
    h.reset()
    for t in range(4):
        _, _, done, info = h.step(torch.zeros((128, 3, 2), device=DEV))
        assert done.dtype == torch.bool and done.data_ptr() == h._done.data_ptr()
