"""GPU tests (-m gpu): the device policies (madrl_heuristic_*) against golden outputs of the unmodified reference
policies and, on live batched envs, against the NumPy oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
G = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "heuristics.npz"))


def test_pursuit_policy_matches_reference_golden_both_layouts():
    from madrl_amd.heuristics import PursuitHeuristicPolicy
    for R in (7, 5, 11):
        win, ref = G["pursuit_R%d_obs" % R], G["pursuit_R%d_act" % R]
        B = len(win)
        det = ref >= 0
        # (R, R, 4) windows, as PursuitEvade(flatten=False) returns them
        a = PursuitHeuristicPolicy(R, flatten=False, seed=5)(torch.as_tensor(win, device=DEV).view(B, 1, R, R, 4)).cpu().numpy()[:, 0]
        assert np.array_equal(a[det], ref[det]), R
        rnd = a[~det]
        assert rnd.min() >= 0 and rnd.max() <= 4 and len(np.unique(rnd)) == 5
        # flatten rows: [ch][x][y] of channels 0..2, then the id
        rows = np.concatenate([np.transpose(win[..., :3], (0, 3, 1, 2)).reshape(B, -1), np.full((B, 1), 0.25, np.float32)], axis=1)
        pol = PursuitHeuristicPolicy(R, flatten=True, seed=5)
        a2 = pol(torch.as_tensor(rows, device=DEV).view(B // 1, 1, -1)).cpu().numpy()[:, 0]
        assert np.array_equal(a2[det], ref[det]), R
        assert np.array_equal(a2[~det], rnd)            # same seed, same row ids, same tick -> same draws
        a3 = pol(torch.as_tensor(rows, device=DEV).view(B, 1, -1)).cpu().numpy()[:, 0]
        assert not np.array_equal(a3[~det], rnd)        # the tick advances


def test_waterworld_and_multiwalker_policies_match_reference_golden():
    from madrl_amd.heuristics import WaterworldHeuristicPolicy, MultiWalkerHeuristicPolicy
    o = torch.as_tensor(G["waterworld_obs"], device=DEV)
    a = WaterworldHeuristicPolicy()(o.view(50, 31, -1)).cpu().numpy().reshape(-1, 2)
    assert np.abs(a - G["waterworld_act"]).max() < 1e-5
    o = torch.as_tensor(G["multiwalker_obs"], device=DEV)
    a = MultiWalkerHeuristicPolicy()(o.view(1000, 3, -1)).cpu().numpy().reshape(-1, 4)
    assert np.abs(a - G["multiwalker_act"]).max() < 1e-6


def test_heuristic_rollouts_on_live_envs():
    from madrl_amd.maps import rectangle_map
    from madrl_amd.pursuit import BatchedPursuitEvade
    from madrl_amd.waterworld import BatchedMAWaterWorld
    from madrl_amd.multiwalker import BatchedMultiWalkerEnv
    from madrl_amd.heuristics import PursuitHeuristicPolicy, WaterworldHeuristicPolicy, MultiWalkerHeuristicPolicy
    from madrl_amd.rollout import RolloutCollector
    from oracle import heuristics_oracle as ho
    # the reference's own evaluation setup (heuristics/pursuit.py:64-67): 16x16, 8 v 30, obs_range 7, n_catch 4, no surround
    N, T = 256, 120
    mk = lambda: BatchedPursuitEvade([rectangle_map(16, 16)], n_envs=N, device=DEV, seed=2, n_pursuers=8, n_evaders=30, obs_range=7,
                                     n_catch=4, surround=False, flatten=False, max_steps=500, auto_reset=True)
    env = mk()
    pol = PursuitHeuristicPolicy(7, flatten=False, seed=9)
    obs = env.reset()
    act = pol(obs)
    ref = ho.pursuit_actions(obs.cpu().numpy().reshape(-1, 7, 7, 4))
    a = act.cpu().numpy().reshape(-1)
    assert np.array_equal(a[ref >= 0], ref[ref >= 0]) and (ref >= 0).mean() > 0.3
    chase = RolloutCollector(env, pol, T).collect().rewards.sum().item()
    rnd = RolloutCollector(mk(), lambda o: torch.randint(0, 5, (N, 8), device=DEV, dtype=torch.int32), T).collect().rewards.sum().item()
    assert chase > 1.5 * rnd, (chase, rnd)            # chasing evaders beats random walking
    # waterworld / multiwalker: device policy == oracle on live observations, rollout runs
    ww = BatchedMAWaterWorld(5, 10, n_envs=128, device=DEV, seed=1)
    wp = WaterworldHeuristicPolicy()
    o = ww.reset()
    assert np.abs(wp(o).cpu().numpy().reshape(-1, 2) - ho.waterworld_actions(o.cpu().numpy().reshape(-1, o.shape[-1]))).max() < 1e-5
    tr = RolloutCollector(ww, wp, 30).collect()
    assert torch.isfinite(tr.rewards).all()
    mw = BatchedMultiWalkerEnv(n_walkers=3, n_envs=64, device=DEV, seed=1)
    mp = MultiWalkerHeuristicPolicy()
    o = mw.reset()
    assert np.abs(mp(o).cpu().numpy().reshape(-1, 4) - ho.multiwalker_actions(o.cpu().numpy().reshape(-1, o.shape[-1]))).max() < 1e-6
    tr = RolloutCollector(mw, mp, 30).collect()
    assert torch.isfinite(tr.rewards).all()


@pytest.mark.parametrize("n_rows", [7, 16, 1000, 70000, 300001])
def test_pursuit_policy_rows_kernel_any_row_count_and_its_own_draw_counter(n_rows):
    """Flatten rows go through the 16-bytes-per-lane kernel (two rows per lane group in flight, the last lanes stepping back to the final
    four cells): every row count -- fewer rows than lane groups, odd tails, more rows than one sweep of the grid -- against the NumPy
    oracle; the launch advances the draw counter itself and leaves its workgroup counts at zero."""
    from madrl_amd import _lib
    from madrl_amd.heuristics import PursuitHeuristicPolicy
    from oracle import heuristics_oracle as ho
    R = 7
    rng = np.random.RandomState(n_rows)
    win = np.zeros((n_rows, R, R, 4), np.float32)
    win[..., 2] = (rng.rand(n_rows, R, R) < 0.04) * rng.randint(1, 4, (n_rows, R, R))      # sparse evader counts, many empty windows
    win[..., 0] = rng.rand(n_rows, R, R) < 0.2
    win[..., 1] = rng.randint(0, 3, (n_rows, R, R))
    ref = ho.pursuit_actions(win[:20000]) if n_rows > 20000 else ho.pursuit_actions(win)
    rows = np.concatenate([np.transpose(win[..., :3], (0, 3, 1, 2)).reshape(n_rows, -1), np.full((n_rows, 1), 0.5, np.float32)], axis=1)
    pol = PursuitHeuristicPolicy(R, flatten=True, seed=11)
    obs = torch.as_tensor(rows, device=DEV).view(n_rows, 1, -1)
    a = pol(obs).cpu().numpy()[:, 0]
    det = ref >= 0
    assert np.array_equal(a[:len(ref)][det], ref[det])
    # the (R, R, 4) layout takes the generic kernel: same actions everywhere, the drawn ones included (same seed, row ids and tick)
    b = PursuitHeuristicPolicy(R, flatten=False, seed=11)(torch.as_tensor(win, device=DEV).view(n_rows, 1, R, R, 4)).cpu().numpy()[:, 0]
    assert np.array_equal(a, b)
    for _ in range(4):
        pol(obs)
    torch.cuda.synchronize()
    t = pol._tick.cpu().numpy()
    assert len(t) == _lib.POLICY_COUNTER_WORDS and t[0] == 5 and not t[1:].any()
