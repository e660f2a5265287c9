"""Rollout collector + discounted-return/GAE scan.  CPU: the NumPy oracle against scipy.signal.lfilter (the
published discount_cumsum the reference's samplers use).  GPU: madrl_rollout_gae and RolloutCollector against
the oracle, on a live auto-reset Pursuit batch."""
import numpy as np
import pytest
import scipy.signal

from oracle import rollout_oracle as ro


def discount_cumsum(x, d):
    return scipy.signal.lfilter([1], [1, float(-d)], x[::-1], axis=0)[::-1]


def test_gae_oracle_matches_lfilter_per_episode():
    rng = np.random.RandomState(0)
    T, N, A, g, lam = 60, 7, 3, 0.97, 0.9
    rew = rng.randn(T, N, A).astype(np.float32)
    val = rng.randn(T + 1, N, A).astype(np.float32)
    done = (rng.rand(T, N) < 0.1).astype(np.uint8) * rng.randint(1, 4, (T, N)).astype(np.uint8)
    done[-1, :3] = 1
    ret, adv = ro.gae(rew, done, val, g, lam)
    ret0, _ = ro.gae(rew, done, None, g, lam)
    for n in range(N):
        cuts = [t + 1 for t in range(T) if done[t, n]]
        ended = bool(cuts) and cuts[-1] == T
        if not ended:
            cuts.append(T)
        s = 0
        for k, e in enumerate(cuts):
            last = (k == len(cuts) - 1) and not ended
            for a in range(A):
                r = rew[s:e, n, a].astype(np.float64)
                v = val[s:e + 1, n, a].astype(np.float64).copy()
                if not last:
                    v[-1] = 0.0                       # terminal: no bootstrap
                r_boot = r.copy(); r_boot[-1] += g * v[-1]
                assert np.allclose(ret[s:e, n, a], discount_cumsum(r_boot, g), atol=1e-10)
                assert np.allclose(ret0[s:e, n, a], discount_cumsum(r, g), atol=1e-10)
                deltas = r + g * v[1:] - v[:-1]
                assert np.allclose(adv[s:e, n, a], discount_cumsum(deltas, g * lam), atol=1e-10)
            s = e


@pytest.mark.gpu
def test_gae_kernel_matches_oracle():
    import torch
    from madrl_amd import _lib
    rng = np.random.RandomState(1)
    for (T, N, A, use_v) in ((1, 1, 1, True), (50, 300, 8, True), (33, 1000, 5, False), (200, 64, 3, True)):
        rew = rng.randn(T, N, A).astype(np.float32)
        val = rng.randn(T + 1, N, A).astype(np.float32) if use_v else None
        done = ((rng.rand(T, N) < 0.05) * rng.randint(1, 4, (T, N))).astype(np.uint8)
        d = lambda x: torch.as_tensor(x, device="cuda:0")
        r_d, dn_d = d(rew), d(done)
        v_d = d(val) if use_v else None
        ret_d = torch.empty_like(r_d); adv_d = torch.empty_like(r_d) if use_v else None
        _lib.check(_lib.lib().madrl_rollout_gae(_lib.ptr(r_d), _lib.ptr(dn_d), _lib.ptr(v_d) if use_v else None, T, N, A, 0.99, 0.95,
                                                _lib.ptr(ret_d), _lib.ptr(adv_d) if use_v else None, _lib.current_stream(r_d.device)))
        ret, adv = ro.gae(rew, done, val, 0.99, 0.95)
        assert np.abs(ret_d.cpu().numpy() - ret).max() < 1e-5 * max(1, np.abs(ret).max())
        if use_v:
            assert np.abs(adv_d.cpu().numpy() - adv).max() < 1e-5 * max(1, np.abs(adv).max())
    with pytest.raises(_lib.MadrlError):
        _lib.check(_lib.lib().madrl_rollout_gae(_lib.ptr(r_d), _lib.ptr(dn_d), None, T, N, A, 0.99, 0.95, _lib.ptr(ret_d), _lib.ptr(ret_d), None))


@pytest.mark.gpu
def test_collector_on_live_pursuit_matches_stepwise_replay():
    import torch
    from madrl_amd.maps import rectangle_map
    from madrl_amd.pursuit import BatchedPursuitEvade
    from madrl_amd.rollout import RolloutCollector
    N, P, T = 256, 8, 40
    mk = lambda: BatchedPursuitEvade([rectangle_map(16, 16)], n_envs=N, device="cuda:0", seed=11, max_steps=12, auto_reset=True,
                                     n_pursuers=P, n_evaders=30, obs_range=7, reward_mech="local")

    def policy(obs):  # deterministic in the observation: replayable
        s = obs.sum(dim=2)
        return (s.abs() * 7.3).to(torch.int32) % 5, torch.tanh(s * 0.01)

    col = RolloutCollector(mk(), policy, T, discount=0.95, gae_lambda=0.8, store_observations=True)
    for it in range(2):   # the second call continues the same episodes
        tr = col.collect()
        if it == 0:
            raw = mk(); obs = raw.reset()
        R, Dn, V, O = [], [], [], []
        for t in range(T):
            a, v = policy(obs)
            O.append(obs.cpu().numpy().copy()); V.append(v.cpu().numpy())
            obs, r, dn, info = raw.step(a)
            R.append(r.cpu().numpy()); Dn.append(info["done_bits"].cpu().numpy())
        V.append(policy(obs)[1].cpu().numpy())
        R, Dn, V = np.stack(R), np.stack(Dn), np.stack(V)
        assert np.array_equal(tr.observations.cpu().numpy(), np.stack(O))
        assert np.array_equal(tr.rewards.cpu().numpy(), R) and np.array_equal(tr.dones.cpu().numpy(), Dn)
        assert (Dn != 0).sum() >= 3 * N
        ret, adv = ro.gae(R, Dn, V, 0.95, 0.8)
        assert np.abs(tr.returns.cpu().numpy() - ret).max() < 1e-5 and np.abs(tr.advantages.cpu().numpy() - adv).max() < 1e-5
    paths = tr.paths(env_ids=[0, 5])
    assert sum(len(p["rewards"]) for p in paths) == 2 * P * T
    p0 = paths[0]
    assert np.allclose(p0["returns"][-1], p0["rewards"][-1] if p0["terminated"] else p0["returns"][-1])
    assert set(p0) >= {"observations", "actions", "rewards", "returns", "advantages", "env_id", "agent_id"}


@pytest.mark.gpu
def test_collector_hipgraph_replay_matches_eager():
    import time
    import torch
    from madrl_amd.maps import rectangle_map
    from madrl_amd.pursuit import BatchedPursuitEvade
    from madrl_amd.heuristics import PursuitHeuristicPolicy
    from madrl_amd.rollout import RolloutCollector
    N, T = 512, 25
    mk = lambda: BatchedPursuitEvade([rectangle_map(16, 16)], n_envs=N, device="cuda:0", seed=4, max_steps=20, auto_reset=True,
                                     n_pursuers=8, n_evaders=30, obs_range=7, n_catch=2, surround=True)
    cols = [RolloutCollector(mk(), PursuitHeuristicPolicy(7, seed=3), T, graph=g, store_observations=True) for g in (False, True)]
    for it in range(4):      # call 1 eager warm-up, call 2 captures + replays, calls 3-4 replay
        a, b = cols[0].collect(), cols[1].collect()
        for k in ("actions", "rewards", "dones", "returns", "observations"):
            assert torch.equal(getattr(a, k), getattr(b, k)), (it, k)
    assert cols[1]._graph is not None
    assert (a.dones != 0).sum() > N
    dt = []
    for c in cols:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            c.collect()
        torch.cuda.synchronize(); dt.append(time.perf_counter() - t0)
    print("collector %d envs x %d steps: eager %.2f ms, hipGraph %.2f ms per collect" % (N, T, dt[0] * 200, dt[1] * 200))
