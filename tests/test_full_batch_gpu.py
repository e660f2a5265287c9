"""Full-batch parity at the sizes BASELINE.json quotes (configs[1], [2], [4]): the HIP kernels against the CPU oracles on EVERY
env of the batch, free-running with fused auto-resets inside the compared region.

  * configs[1]  PursuitEvade 16x16, 8 v 30, 65 536 envs            pursuit_wave_kernel      bit-exact
  * the same at 98 304 envs: more than 375 MB per launch, so successive launches really walk the env range in opposite
    directions (pursuit.hip `launch`)                                                          bit-exact
  * configs[4]  32x32, 16 v 60, one GPU's shard of 32 768 envs      pursuit_group_kernel     bit-exact
  * the authors' training shape (runners/old/rllab/pursuit.sh:1): 32x32 pool, 30 v 50, obs_range 11, sample_maps, 16 384 envs
                                                                    pursuit_group_kernel, LDS slot table   bit-exact
  * configs[2]  MAWaterWorld 5 / 10 / 30 sensors, 32 768 envs       waterworld_kernel<1,5,10,10,30>  == the float32 oracle

The oracles are the C restatements pinned to the unmodified reference by tests/test_oracle_*.py; the reference's own draws
cannot be replayed at this size, so both sides run the Philox contract of DESIGN.md.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

PURSUIT = {
    # name: (map side, P, E, envs, steps, horizon)
    "c2_65536": (16, 8, 30, 65536, 30, 11),
    "c2_98304_alternating_walk": (16, 8, 30, 98304, 14, 6),
    "c5_shard_32768": (32, 16, 60, 32768, 16, 7),
    "authors_30v50_obs11_16384": ("pool32", 30, 50, 16384, 12, 5),
}


@pytest.mark.parametrize("case", sorted(PURSUIT), ids=sorted(PURSUIT))
def test_pursuit_full_batch_bit_exact(case):
    from madrl_amd.maps import rectangle_map
    from madrl_amd.pursuit import BatchedPursuitEvade
    from oracle import pursuit as po
    side, P, E, N, T, H = PURSUIT[case]
    if side == "pool32":   # TwoDMaps.resize(2, map_pool16) as recorded in the golden of this shape; sample_maps like the authors' launch line
        import os
        from conftest import GOLDEN
        maps = list(np.load(os.path.join(GOLDEN, "pursuit_authors_30v50_obs11.npz"))["maps"])
        kw = dict(n_pursuers=P, n_evaders=E, obs_range=11, n_catch=2, surround=True, flatten=True, reward_mech="local", sample_maps=True)
    else:
        maps = [rectangle_map(side, side)]
        kw = dict(n_pursuers=P, n_evaders=E, obs_range=7, n_catch=2, surround=True, flatten=True, reward_mech="local")
    env = BatchedPursuitEvade(maps, n_envs=N, device=DEV, seed=99, env_id_base=7, max_steps=H, auto_reset=True, **kw)
    assert env.kernel_kind == "wave"
    orc = po.PursuitOracle(maps, n_envs=N, seed=99, env_id_base=7, **kw)
    obs = env.reset()
    assert np.array_equal(obs.cpu().numpy(), orc.reset()), "reset observations"
    # episode ages spread over [0, H): every launch carries its share of fused resets, as in a steady rollout
    age = (np.arange(N) % H).astype(np.int32)
    env.set_state(dict(t=age))
    tstep = age.astype(np.int64).copy()
    rng = np.random.RandomState(17)
    n_resets = n_removed = 0
    for t in range(T):
        act = rng.randint(5, size=(N, P)).astype(np.int32)
        obs, rew, done, info = env.step(torch.as_tensor(act, device=DEV))
        oobs, orew, odone, orem = orc.step(act)
        tstep += 1
        bits = odone.astype(np.uint8) | ((tstep >= H).astype(np.uint8) << 1)
        assert np.array_equal(info["done_bits"].cpu().numpy(), bits), "step %d done bits" % t
        assert np.array_equal(info["removed"].cpu().numpy(), orem), "step %d removed" % t
        assert np.array_equal(rew.cpu().numpy(), orew.astype(np.float32)), "step %d rewards" % t
        mask = (bits != 0).astype(np.uint8)
        if mask.any():
            orc.reset(mask=mask)
            tstep[mask != 0] = 0
        n_resets += int(mask.sum())
        n_removed += int(orem.sum())
        got = obs.cpu().numpy()
        assert np.array_equal(got, orc.obs), "step %d: %d observation cells differ" % (t, int((got != orc.obs).sum()))
    gst, ost = env.get_state(), orc.get_state()
    for k in ("pos_p", "pos_e", "gone", "term_p", "term_e", "map_id"):
        assert np.array_equal(gst[k].cpu().numpy(), ost[k]), "final state[%s]" % k
    assert np.array_equal(gst["tick"].cpu().numpy().view(np.uint32), ost["tick"])
    assert np.array_equal(gst["t"].cpu().numpy(), tstep)
    assert n_resets >= N and n_removed > 0, (n_resets, n_removed)   # every env went through the fused reset at least once


@pytest.mark.parametrize("mode", ["headline", "secondary_hwc", "authors_long_rows"])
@pytest.mark.parametrize("start", ["declared_zero", "nothing_known"])
def test_long_rollout_into_the_mask_equilibrium_is_bit_exact(mode, start):
    """The fast path decides per 16-byte slot between one whole store (cells outside the map that are KNOWN to hold 0.0 are written as
    zeros, and all-outside slots complete their 64-byte chunk) and masked 4-byte stores (a stale value must survive), from the stale-zero
    masks it keeps.  What is known changes for ~2 000 steps; this runs 2 048 envs for 1 200 steps from both starting states of the masks --
    the buffer declared all-zero (what the Python layer does for the buffer it allocates) and "nothing known" (invalidate_obs) -- and
    compares the WHOLE persistent observation buffer with the oracle's every 100 steps."""
    from madrl_amd.maps import rectangle_map
    from madrl_amd.pursuit import BatchedPursuitEvade
    from oracle import pursuit as po
    N, P, E, T, H = 2048, 8, 30, 1200, 97
    maps = [rectangle_map(16, 16)]
    kw = dict(n_pursuers=P, n_evaders=E, obs_range=7, reward_mech="local")
    kw.update(dict(n_catch=2, surround=True, flatten=True) if mode == "headline" else dict(n_catch=2, surround=False, flatten=False))
    if mode == "authors_long_rows":   # the two-wavefront kernel with the LDS slot table: three stale-zero mask words per thread
        import os
        from conftest import GOLDEN
        N, P, E, T = 192, 30, 50, 500
        maps = list(np.load(os.path.join(GOLDEN, "pursuit_authors_30v50_obs11.npz"))["maps"])
        kw = dict(n_pursuers=P, n_evaders=E, obs_range=11, n_catch=2, surround=True, flatten=True, reward_mech="local", sample_maps=True)
    env = BatchedPursuitEvade(maps, n_envs=N, device=DEV, seed=5, env_id_base=1 << 20, max_steps=H, auto_reset=True, **kw)
    assert env.kernel_kind == "wave"
    if start == "nothing_known":
        env.invalidate_obs()
    orc = po.PursuitOracle(maps, n_envs=N, seed=5, env_id_base=1 << 20, **kw)
    assert np.array_equal(env.reset().cpu().numpy().reshape(N, P, -1), orc.reset())
    rng = np.random.RandomState(3)
    tstep = np.zeros(N, np.int64)
    for t in range(T):
        act = rng.randint(5, size=(N, P)).astype(np.int32)
        obs, rew, done, info = env.step(torch.as_tensor(act, device=DEV))
        oobs, orew, odone, orem = orc.step(act)
        tstep += 1
        mask = (odone.astype(np.uint8) | ((tstep >= H).astype(np.uint8) << 1)) != 0
        if mask.any():
            orc.reset(mask=mask.astype(np.uint8))
            tstep[mask] = 0
        if t % 100 == 99 or t < 3:
            got = obs.cpu().numpy().reshape(N, P, -1)
            assert np.array_equal(got, orc.obs), "step %d: %d observation cells differ" % (t, int((got != orc.obs).sum()))
            assert np.array_equal(rew.cpu().numpy(), orew.astype(np.float32)) and np.array_equal(info["removed"].cpu().numpy(), orem)


def test_waterworld_c3_full_batch_matches_f32_oracle():
    """32 768 envs, free-running (no re-synchronisation), auto-reset at a short horizon so that the respawn / reset paths
    run inside the compared region: every output of every step equals the float32 oracle's bit for bit."""
    from madrl_amd.waterworld import BatchedMAWaterWorld
    from oracle import waterworld as ww
    N, T, H = 32768, 24, 9
    env = BatchedMAWaterWorld(5, 10, n_envs=N, device=DEV, seed=31, env_id_base=11, max_steps=H, auto_reset=True)
    orc = ww.WaterworldOracle(5, 10, n_envs=N, seed=31, env_id_base=11, max_steps=H, dtype=np.float32)
    assert np.array_equal(env.reset().cpu().numpy(), orc.reset()), "reset observations"
    rng = np.random.RandomState(3)
    catches = 0
    for t in range(T):
        act = rng.uniform(-1, 1, size=(N, 5, 2)).astype(np.float32)
        obs, rew, done, info = env.step(act)
        oobs, orew, odone, oinfo = orc.step(act)
        assert np.array_equal(done.cpu().numpy(), odone.astype(bool)), "done step %d" % t
        assert np.array_equal(info["evcatches"].cpu().numpy(), oinfo[:, 0]) and np.array_equal(info["pocatches"].cpu().numpy(), oinfo[:, 1])
        assert np.array_equal(rew.cpu().numpy(), orew), "rewards step %d" % t
        catches += int(oinfo.sum())
        if odone.any():
            orc.reset(mask=odone)
        got = obs.cpu().numpy()
        assert np.array_equal(got, orc.obs), "obs step %d: max |d| = %g" % (t, np.abs(got - orc.obs).max())
    gst, ost = env.get_state(), orc.get_state()
    assert np.array_equal(gst["pos"].cpu().numpy(), ost["pos"]) and np.array_equal(gst["vel"].cpu().numpy(), ost["vel"])
    assert np.array_equal(gst["t"].cpu().numpy(), ost["t"])
    assert catches > 0
