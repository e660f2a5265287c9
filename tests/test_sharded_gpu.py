"""GPU tests (-m gpu): one env batch stepped as sub-batches on their own HIP streams (madrl_amd/sharded.py) gives exactly the results of
the same batch stepped by one launch -- sub-batch j carries env ids [j * per, (j + 1) * per)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_pursuit_sub_batches_on_streams_equal_one_batch():
    from madrl_amd.maps import rectangle_map
    from madrl_amd.pursuit import BatchedPursuitEvade
    from madrl_amd.sharded import StreamSharded
    kw = dict(n_pursuers=8, n_evaders=30, obs_range=7, n_catch=2, surround=True, flatten=True, seed=7, max_steps=25, auto_reset=True)
    maps = [rectangle_map(16, 16)]
    N = 4096
    one = BatchedPursuitEvade(maps, n_envs=N, device=DEV, env_id_base=100, **kw)
    sh = StreamSharded(lambda n_envs, env_id_base, device: BatchedPursuitEvade(maps, n_envs=n_envs, device=device, env_id_base=env_id_base, **kw),
                       N, n_streams=4, env_id_base=100, device=DEV)
    o1 = one.reset()
    o4 = sh.reset()
    assert torch.equal(o1, torch.cat(o4))
    g = torch.Generator(device="cpu").manual_seed(1)
    for t in range(60):   # crosses the horizon twice: fused resets inside the compared region
        a = torch.randint(0, 5, (N, 8), generator=g, dtype=torch.int32).to(DEV)
        o1, r1, d1, i1 = one.step(a)
        parts = sh.step(a, join=(t % 3 == 0))     # free-running between joins
        if t % 3 == 0:
            assert torch.equal(o1, torch.cat([p[0] for p in parts])) and torch.equal(r1, torch.cat([p[1] for p in parts]))
            assert torch.equal(d1, torch.cat([p[2] for p in parts])) and torch.equal(i1["removed"], torch.cat([p[3]["removed"] for p in parts]))
    sh.join()
    torch.cuda.synchronize()
    assert torch.equal(one._obs, torch.cat([e._obs for e in sh.envs]))


def test_standardized_waterworld_sub_batches_equal_one_batch():
    from madrl_amd.waterworld import BatchedMAWaterWorld
    from madrl_amd.wrappers import StandardizedEnv
    from madrl_amd.sharded import StreamSharded
    N = 1024
    mk = lambda n_envs, env_id_base, device, fused=None: StandardizedEnv(BatchedMAWaterWorld(5, 10, n_envs=n_envs, device=device, seed=3, env_id_base=env_id_base, auto_reset=True),
                                                             scale_reward=0.5, enable_obsnorm=True, enable_rewnorm=True, fused=fused)
    for fused in (None, False):   # the fused epilogue and the stand-alone kernels (16-byte pair kernel)
        one = mk(N, 0, DEV, fused)
        sh = StreamSharded(lambda n_envs, env_id_base, device: mk(n_envs, env_id_base, device, fused), N, n_streams=2, device=DEV)
        a0 = one.reset().clone()
        assert torch.equal(a0, torch.cat(sh.reset()))
        g = torch.Generator(device="cpu").manual_seed(2)
        for t in range(12):
            act = (torch.rand((N, 5, 2), generator=g) * 2 - 1).to(DEV)
            o1, r1, d1, _ = one.step(act)
            parts = sh.step(act)
            assert torch.equal(o1, torch.cat([p[0] for p in parts])), (fused, t)
            assert torch.equal(r1, torch.cat([p[1] for p in parts])) and torch.equal(d1, torch.cat([p[2] for p in parts]))


def test_sharded_rollout_collector_equals_the_plain_collector():
    """policy in the loop: two sub-batches driven on their own streams, steps interleaved, give the trajectory of the one-batch collector"""
    from madrl_amd.maps import rectangle_map
    from madrl_amd.pursuit import BatchedPursuitEvade
    from madrl_amd.rollout import RolloutCollector, ShardedRolloutCollector
    from madrl_amd.sharded import StreamSharded
    N, P, T = 512, 8, 30
    mk = lambda n_envs=N, env_id_base=0, device=DEV: BatchedPursuitEvade([rectangle_map(16, 16)], n_envs=n_envs, device=device, seed=11, env_id_base=env_id_base,
                                                                         max_steps=12, auto_reset=True, n_pursuers=P, n_evaders=30, obs_range=7, reward_mech="local")

    def policy(obs):  # deterministic in the observation
        s = obs.sum(dim=2)
        return (s.abs() * 7.3).to(torch.int32) % 5, torch.tanh(s * 0.01)

    one = RolloutCollector(mk(), policy, T, discount=0.95, gae_lambda=0.8, store_observations=True)
    two = ShardedRolloutCollector(StreamSharded(mk, N, n_streams=2, device=DEV), [policy, policy], T, discount=0.95, gae_lambda=0.8, store_observations=True)
    graphed = ShardedRolloutCollector(StreamSharded(mk, N, n_streams=2, device=DEV), [policy, policy], T, discount=0.95, gae_lambda=0.8, store_observations=True,
                                      graph=True)   # one hipGraph per sub-batch and horizon, replayed on the sub-batch's stream
    for it in range(3):
        a = one.collect()
        for col in (two, graphed):
            parts = col.collect()
            torch.cuda.synchronize()
            for k in ("observations", "actions", "rewards", "dones", "returns", "advantages", "values"):
                assert torch.equal(getattr(a, k), torch.cat([getattr(p, k) for p in parts], dim=1)), (it, k, col is graphed)
