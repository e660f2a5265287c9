"""GPU tests (-m gpu): edge cases of the particle envs -- maximum particle counts (62 per env: one wavefront), many sensors,
tiny and ragged batches, configurations the library must refuse -- against the float32 oracles."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run_ww(kw, N, T, seed):
    from madrl_amd.waterworld import BatchedMAWaterWorld
    from oracle import waterworld as ww
    env = BatchedMAWaterWorld(n_envs=N, device=DEV, seed=seed, max_steps=15, auto_reset=False, **kw)
    orc = ww.WaterworldOracle(n_envs=N, seed=seed, max_steps=15, dtype=np.float32, **kw)
    assert np.array_equal(env.reset().cpu().numpy(), orc.reset())
    rng = np.random.RandomState(seed)
    Np = kw["n_pursuers"]
    for t in range(T):
        a = rng.uniform(-1, 1, (N, Np, 2)).astype(np.float32)
        o, r, d, info = env.step(a)
        oo, orr, od, oi = orc.step(a)
        assert np.array_equal(o.cpu().numpy(), oo) and np.array_equal(r.cpu().numpy(), orr), t
        assert np.array_equal(d.cpu().numpy(), od != 0) and np.array_equal(info["evcatches"].cpu().numpy(), oi[:, 0])
        if od.any():
            m = od != 0
            assert np.array_equal(env.reset(mask=m).cpu().numpy()[m], orc.reset(mask=m.astype(np.uint8))[m])


def test_waterworld_maximum_particles_and_ragged_batches():
    big = dict(n_pursuers=12, n_evaders=25, n_poison=25, n_coop=3, n_sensors=16, radius=0.03, ev_speed=0.03, action_scale=0.03)  # 62 particles
    _run_ww(big, N=65, T=40, seed=1)           # 65 envs: one more than a wavefront's worth of workgroups per pass
    _run_ww(dict(n_pursuers=1, n_evaders=1, n_poison=1, n_coop=1, n_sensors=1), N=1, T=40, seed=2)   # the smallest env there is
    _run_ww(dict(n_pursuers=2, n_evaders=3, n_poison=2, n_coop=2, n_sensors=200, sensor_range=0.5), N=3, T=30, seed=3)  # many sensors: 7 chunks of pairs


def test_hostage_maximum_particles_and_tiny_batch():
    from madrl_amd.hostage import BatchedContinuousHostageWorld
    from oracle import hostage as ho
    for (args, N, kw) in (((10, 30, 21, 3, 2), 33, dict(n_sensors=8, action_scale=0.03, bad_speed=0.03, radius=0.03)),
                          ((1, 1, 1, 1, 1), 1, dict(n_sensors=2))):
        env = BatchedContinuousHostageWorld(*args, n_envs=N, device=DEV, seed=4, max_steps=20, **kw)
        orc = ho.HostageOracle(*args, n_envs=N, seed=4, max_steps=20, dtype=np.float32, **kw)
        assert np.array_equal(env.reset().cpu().numpy(), orc.reset())
        rng = np.random.RandomState(0)
        for t in range(45):
            a = rng.uniform(-1, 1, (N, args[0], 2)).astype(np.float32)
            o, r, d, info = env.step(a)
            oo, orr, od, oi = orc.step(a)
            assert np.array_equal(o.cpu().numpy(), oo) and np.array_equal(r.cpu().numpy(), orr), t
            assert np.array_equal(d.cpu().numpy(), od != 0)
            if od.any():
                m = od != 0
                assert np.array_equal(env.reset(mask=m).cpu().numpy()[m], orc.reset(mask=m.astype(np.uint8))[m])


def test_configurations_the_library_refuses():
    from madrl_amd import _lib
    from madrl_amd.waterworld import BatchedMAWaterWorld
    from madrl_amd.hostage import BatchedContinuousHostageWorld
    from madrl_amd.multiwalker import BatchedMultiWalkerEnv
    with pytest.raises(_lib.MadrlError, match="62 particles"):
        BatchedMAWaterWorld(13, 25, n_poison=25, n_envs=4, device=DEV)
    with pytest.raises(_lib.MadrlError, match="LDS"):
        BatchedMAWaterWorld(12, 25, n_poison=25, n_sensors=256, n_envs=4, device=DEV)   # 12 x 1795 floats of observation staging
    with pytest.raises(_lib.MadrlError, match="n_sensors"):
        BatchedMAWaterWorld(2, 2, n_sensors=300, n_envs=4, device=DEV)
    with pytest.raises(_lib.MadrlError, match="61 particles"):
        BatchedContinuousHostageWorld(10, 30, 22, 2, 2, n_envs=4, device=DEV)
    for w in (0, 11):   # 1 .. 10 run (the reference's curriculum, lessons/multiwalker/env.yaml)
        with pytest.raises(_lib.MadrlError, match="n_walkers"):
            BatchedMultiWalkerEnv(n_walkers=w, n_envs=4, device=DEV)
    with pytest.raises(_lib.MadrlError, match="no CPU path"):
        BatchedMAWaterWorld(2, 2, n_envs=4, device="cpu")
    env = BatchedMAWaterWorld(2, 2, n_envs=4, device=DEV)
    with pytest.raises(AssertionError):
        env.step(torch.zeros(4, 3, 2, device=DEV))      # wrong action shape (waterworld.py:227)


def test_step_runs_on_the_callers_stream():
    """all launches go to torch's current stream: a rollout on a side stream equals the one on the default stream, and a
    consumer on the same stream sees finished outputs without any explicit synchronisation"""
    from madrl_amd.maps import rectangle_map
    from madrl_amd.pursuit import BatchedPursuitEvade
    mk = lambda: BatchedPursuitEvade([rectangle_map(16, 16)], n_envs=3000, device=DEV, seed=8, max_steps=20, auto_reset=True,
                                     n_pursuers=8, n_evaders=30, obs_range=7)
    a, b = mk(), mk()
    acts = [torch.randint(0, 5, (3000, 8), device=DEV, dtype=torch.int32) for _ in range(30)]
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=DEV)
    sums_a, sums_b = [], []
    a.reset()
    for t in range(30):
        o, r, d, _ = a.step(acts[t]); sums_a.append((o.sum(), r.sum(), d.sum()))
    with torch.cuda.stream(side):
        b.reset()
        for t in range(30):
            o, r, d, _ = b.step(acts[t]); sums_b.append((o.sum(), r.sum(), d.sum()))   # consumers enqueued right behind the step
    side.synchronize(); torch.cuda.synchronize()
    for (x1, y1, z1), (x2, y2, z2) in zip(sums_a, sums_b):
        assert float(x1) == float(x2) and float(y1) == float(y2) and int(z1) == int(z2)
    assert torch.equal(a.obs_buffer, b.obs_buffer)


def test_more_than_253_agents_on_one_cell_raise_the_overflow_bit():
    """The generic kernel counts agents per cell in bytes (254 / 255 are padding sentinels).  With up to 1 023 agents of a kind a cell can
    in principle collect more than 253 of them: the env is then marked (bit 7 of the done byte, info['count_overflow']) instead of
    silently wrong, and a reset clears the mark."""
    import torch
    from madrl_amd.pursuit import BatchedPursuitEvade
    N, P, E = 3, 4, 300
    env = BatchedPursuitEvade([np.zeros((8, 8), np.int32)], n_envs=N, device="cuda:0", seed=0, n_pursuers=P, n_evaders=E, obs_range=3,
                              n_catch=4, surround=False)
    pos = np.zeros((N, P + E, 2), np.int32)
    pos[:, :P] = [[7, 7], [7, 6], [6, 7], [6, 6]]
    pos[0, P:] = [2, 2]                                    # env 0: all 300 evaders on one cell
    pos[1, P:] = np.stack([np.arange(E) % 8, (np.arange(E) // 8) % 6], 1)   # env 1: spread out (<= 7 per cell)
    pos[2, P:P + 253] = [3, 3]; pos[2, P + 253:] = [0, 0]   # env 2: exactly 253 on one cell -- still inside the byte's range
    env.reset(positions=pos)
    stay = torch.full((N, P), 4, dtype=torch.int32, device="cuda:0")
    _, _, _, info = env.step(stay, evader_actions=np.full((N, E), 4, np.int32))
    assert info["count_overflow"].cpu().numpy().tolist() == [True, False, False]
    env.reset(mask=np.array([1, 0, 0], np.uint8), positions=pos[[1, 1, 1]])
    _, _, _, info = env.step(stay, evader_actions=np.full((N, E), 4, np.int32))
    assert not info["count_overflow"].any()


def test_large_generic_waterworld_batch_says_how_to_get_its_kernel():
    """a shape outside csrc/waterworld_specializations.def runs (about half as fast) on the generic instantiation: a batch of 4 096 envs or more
    says so once per shape, with the command that compiles its kernel; specialised shapes and small batches stay quiet"""
    import warnings
    from madrl_amd.waterworld import BatchedMAWaterWorld
    BatchedMAWaterWorld._hinted.clear()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        BatchedMAWaterWorld(5, 10, n_envs=4096, device=DEV)            # BASELINE shape: specialised
        BatchedMAWaterWorld(4, 7, n_envs=256, device=DEV)              # small batch
        assert not [x for x in w if "generic kernel" in str(x.message)]
        e = BatchedMAWaterWorld(4, 7, n_envs=4096, device=DEV)
        BatchedMAWaterWorld(4, 7, n_envs=4096, device=DEV)             # once per shape
    msgs = [str(x.message) for x in w if "generic kernel" in str(x.message)]
    assert len(msgs) == 1 and "--waterworld-shape 4 7 10 30 213" in msgs[0]
    obs = e.reset()
    assert obs.shape == (4096, 4, 213)
