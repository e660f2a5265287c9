"""CPU tests (-m "not gpu"): the C oracle against the golden vectors produced by the
unmodified reference (oracle/make_golden_pursuit.py), plus properties of its RNG paths."""
import numpy as np
import pytest

from oracle import pursuit as po
from helpers import pursuit_golden_files, golden_id


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 10 rounds
    assert [hex(v) for v in po.philox([0] * 4, [0] * 2)] == ['0x6627e8d5', '0xe169c58d', '0xbc57ac4c', '0x9b00dbd8']
    assert [hex(v) for v in po.philox([0xffffffff] * 4, [0xffffffff] * 2)] == [
        '0x408f276d', '0x41c83b0e', '0xa20bc7c6', '0x6d5451fd']
    assert [hex(v) for v in po.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0])] == [
        '0xd16cfe09', '0x94fdcceb', '0x5001e420', '0x24126ea1']


@pytest.mark.parametrize("path", pursuit_golden_files(), ids=golden_id)
def test_oracle_matches_reference_golden(path):
    g = np.load(path)
    assert float(g["obs_cast_err"]) < 1e-7  # the f64 -> f32 cast of reference obs is benign (Q7)
    o = po.PursuitOracle(list(g["maps"]), n_envs=1, **po.config_from_golden(g))
    for t in range(len(g["op"])):
        if g["op"][t] == 0:
            pos = np.concatenate([g["init_p"][t], g["init_e"][t]])[None]
            obs = o.reset(inj_pos=pos, inj_map=np.array([g["map_id"][t]]))
            np.testing.assert_array_equal(obs[0].reshape(g["obs_f32"][t].shape), g["obs_f32"][t], err_msg="reset obs, op %d" % t)
        else:
            obs, rew, done, rem = o.step(g["act_p"][t][None], g["act_e"][t][None])
            st = o.get_state()
            np.testing.assert_array_equal(obs[0].reshape(g["obs_f32"][t].shape), g["obs_f32"][t], err_msg="obs, op %d" % t)
            np.testing.assert_array_equal(rew[0], g["rew_f64"][t], err_msg="rewards (float64 exact), op %d" % t)
            assert int(done[0]) == int(g["done"][t]) and int(rem[0]) == int(g["removed"][t])
            np.testing.assert_array_equal(st["pos_p"][0], g["pos_p"][t])
            np.testing.assert_array_equal(st["pos_e"][0], g["pos_e"][t])
            np.testing.assert_array_equal(st["gone"][0], g["gone_e"][t])


def test_golden_covers_the_quirks():
    """The fixtures must actually exercise what they claim (catches, dones, stale cells...)."""
    files = {golden_id(p): np.load(p) for p in pursuit_golden_files()}
    assert files["pursuit_c1_surround_local"]["removed"].sum() > 5
    assert files["pursuit_tiny5_dense"]["done"].sum() > 0
    assert files["pursuit_c1_colocate_hwc"]["removed"].sum() > 5
    g = files["pursuit_pool16_sample_maps"]
    assert len(np.unique(g["map_id"])) > 3
    g = files["pursuit_random_opponents"]      # :177-181: the evader count is redrawn by every reset
    created = [(g["init_e"][t][:, 0] >= 0).sum() for t in np.where(g["op"] == 0)[0]]
    assert len(set(created)) >= 4 and min(created) >= 1 and max(created) < int(g["cfg_max_opponents"])
    # Q2: some out-of-map cell of channel 1/2 holds a non-zero (stale) value in a golden obs
    g = files["pursuit_c1_surround_local"]
    R = int(g["cfg_obs_range"])
    obs = g["obs_f32"][:, :, :3 * R * R].reshape(len(g["op"]), -1, 3, R, R)
    pos = g["pos_p"]
    stale = 0
    off = (R - 1) // 2
    for t in range(len(g["op"])):
        for p in range(pos.shape[1]):
            x, y = pos[t, p]
            for i in range(R):
                gx = x - off + i
                if gx < 0 or gx >= int(g["cfg_xs"]):
                    stale += int((obs[t, p, 1:, i, :] != 0).sum())
    assert stale > 0


def test_oracle_reset_is_uniform_over_free_cells():
    from madrl_amd.maps import rectangle_map
    m = rectangle_map(16, 16)
    N = 4096
    o = po.PursuitOracle([m], n_envs=N, seed=123, n_pursuers=8, n_evaders=30, obs_range=7, reward_mech="local")
    o.reset()
    st = o.get_state()
    assert (m[st["pos_p"][..., 0], st["pos_p"][..., 1]] == 0).all()
    assert (m[st["pos_e"][..., 0], st["pos_e"][..., 1]] == 0).all()
    cnt = np.zeros((16, 16))
    np.add.at(cnt, (st["pos_e"][..., 0].ravel(), st["pos_e"][..., 1].ravel()), 1)
    free = m == 0
    exp = cnt.sum() / free.sum()
    chi2 = (((cnt[free] - exp) ** 2) / exp).sum()
    dof = free.sum() - 1
    assert abs(chi2 - dof) < 6 * np.sqrt(2 * dof), (chi2, dof)


def test_oracle_free_running_is_deterministic_and_seed_sensitive():
    from madrl_amd.maps import rectangle_map
    m = rectangle_map(16, 16)
    kw = dict(n_pursuers=8, n_evaders=30, obs_range=7, reward_mech="local")
    outs = []
    for seed in (7, 7, 8):
        o = po.PursuitOracle([m], n_envs=32, seed=seed, **kw)
        o.reset()
        rng = np.random.RandomState(0)
        for _ in range(20):
            o.step(rng.randint(5, size=(32, 8)))
        outs.append(o.get_state()["pos_e"].copy())
    assert np.array_equal(outs[0], outs[1])
    assert not np.array_equal(outs[0], outs[2])


def test_free_running_reset_samples_the_reference_distribution():
    """The free-running reset (own Philox draws) against the UNMODIFIED reference's reset() distribution with a constraint window of
    0.5 and a sampled 10-map pool (tests/golden/resetdist_pursuit.npz, 40 000 reference resets, oracle/make_golden_reset_hist.py).
    Agents of one reset share its window, so only ONE position per reset enters each chi-square (agent indices 0, 7, 8, 37, pooled over
    the maps: independent samples); the map choice is tested the same way; the all-agent histograms are checked for support only.
    The HIP kernels are bit-identical to this oracle in free-running mode (tests/test_pursuit_gpu.py cases `pool16`, `tiny_window`)."""
    import os
    from oracle import pursuit as po
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "resetdist_pursuit.npz"))
    maps = [m.astype(np.int32) for m in g["maps"]]
    P, E = int(g["n_pursuers"]), int(g["n_evaders"])
    N = 40000
    orc = po.PursuitOracle(maps, n_envs=N, seed=7, n_pursuers=P, n_evaders=E, obs_range=7, constraint_window=float(g["constraint_window"]),
                           sample_maps=True)
    orc.reset()
    st = orc.get_state()
    pos = np.concatenate([st["pos_p"], st["pos_e"]], axis=1)   # [N, P + E, 2]
    cell = pos[..., 0].astype(np.int64) * 16 + pos[..., 1]
    nmap = np.bincount(st["map_id"], minlength=len(maps))

    def two_sample_chi2(a, b):
        a, b = a.astype(np.float64), b.astype(np.float64)
        keep = (a + b) >= 10
        A, B = a[keep].sum(), b[keep].sum()
        stat = (((a[keep] * np.sqrt(B / A) - b[keep] * np.sqrt(A / B)) ** 2) / (a[keep] + b[keep])).sum()
        return stat, int(keep.sum()) - 1

    stat, dof = two_sample_chi2(nmap, g["nmap"])
    assert abs(stat - dof) < 6 * np.sqrt(2 * dof), ("map choice", stat, dof)
    for k, agent in enumerate(g["agents"]):
        mine = np.bincount(cell[:, int(agent)], minlength=256)
        stat, dof = two_sample_chi2(mine, g["one"][k])
        assert dof > 150 and abs(stat - dof) < 6 * np.sqrt(2 * dof), ("agent %d" % agent, stat, dof)
    hist = np.zeros_like(g["hist"])
    np.add.at(hist, (np.repeat(st["map_id"], cell.shape[1]), cell.ravel()), 1)
    for m in range(len(maps)):  # same support on every map: nobody inside a building, every free cell reachable
        assert np.array_equal(hist[m] > 0, g["hist"][m] > 0) and (hist[m][maps[m].ravel() == -1] == 0).all()
    # the window makes the distribution non-uniform (cells near the middle are covered by more windows): the test has power
    flat = g["one"].sum(0).astype(np.float64)
    assert flat.max() > 1.5 * flat[flat > 0].min()


from helpers import evadercontrol_golden_files, replay_evadercontrol  # noqa: E402


@pytest.mark.parametrize("path", evadercontrol_golden_files(), ids=golden_id)
def test_oracle_matches_reference_golden_evader_control(path):
    """train_pursuit=False (pursuit_evade.py:105-112, :204-207, :215-224): the actions drive the evaders, the pursuers move by
    pursuer_controller, observations are the evaders' windows, rewards stay the pursuers'."""
    g = np.load(path)
    assert float(g["obs_cast_err"]) < 1e-7
    o = po.PursuitOracle(list(g["maps"]), n_envs=1, **po.config_from_golden(g))
    assert not o.cfg.train_pursuit

    def state():
        st = o.get_state()
        return dict(pos_p=st["pos_p"][0], pos_e=st["pos_e"][0], gone=st["gone"][0])

    def step(aa, ao):
        obs, rew, done, rem = o.step(aa, ao)
        return obs[0], rew[0], done[0], rem[0]

    replay_evadercontrol(g, lambda pos: o.reset(inj_pos=pos)[0], step, state)
