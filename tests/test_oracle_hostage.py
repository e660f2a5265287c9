"""CPU tests (-m "not gpu"): the float64 hostage oracle against golden records of the unmodified reference
(teacher-forced: pre-state + action + respawn uniforms -> post-state and outputs)."""
import glob
import os

import numpy as np
import pytest

from oracle import hostage as ho

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "hostage_*.npz")))


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[8:-4] for f in FILES])
def test_oracle_reproduces_reference_records(path):
    g = np.load(path)
    o = ho.HostageOracle(n_envs=1, sensors=g["sensors"], **ho.kwargs_from_golden(g))
    T = len(g["pre_t"])
    Nr, Nh = o.Nr, o.Nh
    for t in range(T):
        o.set_state(**ho.golden_pre_state(g, t))
        resp = np.where(g["resp"][t] >= 0, g["resp"][t], 0.0)
        obs, rew, done, info = o.step(g["act"][t][None], resp=resp[None])
        st = o.get_state()
        assert np.array_equal(st["pos"][0], g["post_pos"][t]), t
        assert np.array_equal(st["vel"][0], g["post_vel"][t]), t
        assert [(int(st["saved"][0]) >> j) & 1 for j in range(Nh)] == list(g["post_saved"][t]), t
        assert int(st["flags"][0]) & 3 == int(g["post_gate"][t]) | (int(g["post_bombed"][t]) << 1), t
        assert int(st["t"][0]) == int(g["post_t"][t])
        assert np.abs(obs[0] - g["obs"][t]).max() < 1e-12, (t, np.abs(obs[0] - g["obs"][t]).max())   # BLAS dot vs a*b + c*d: last-ulp
        if not g["is_reset_step"][t]:
            assert np.abs(rew[0] - g["rew"][t]).max() < 1e-12, (t, rew[0], g["rew"][t])
            assert int(done[0]) == int(g["done"][t]) and list(info[0]) == list(g["info"][t]), t
    assert (g["resp"][..., 0] >= 0).sum() > 3 or "fuzz" in path   # the hand-written scenarios must exercise respawns; the drawn ones take what comes


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[8:-4] for f in FILES])
def test_float32_oracle_stays_within_tolerance_of_the_reference_records(path):
    """the kernel's arithmetic (the float32 build of the same restatement, which tests/test_hostage_gpu.py requires the kernel to equal bit
    for bit): every recorded step within 1e-5 of the unmodified reference -- except a sensing `<=` decided the other way on a value one rounding
    error from its threshold: none in the hand-written records, one in the drawn ones (fuzz_03, step 92)"""
    g = np.load(path)
    o = ho.HostageOracle(n_envs=1, sensors=g["sensors"], dtype=np.float32, **ho.kwargs_from_golden(g))
    flips = []
    for t in range(len(g["pre_t"])):
        o.set_state(**ho.golden_pre_state(g, t))
        resp = np.where(g["resp"][t] >= 0, g["resp"][t], 0.0)
        obs, rew, done, info = o.step(g["act"][t][None], resp=resp[None])
        st = o.get_state()
        assert np.abs(st["pos"][0] - g["post_pos"][t]).max() < 1e-6 and np.abs(st["vel"][0] - g["post_vel"][t]).max() < 1e-6, t
        if np.abs(obs[0] - g["obs"][t]).max() >= 1e-5:
            flips.append(t)
        if not g["is_reset_step"][t]:
            assert np.abs(rew[0] - g["rew"][t]).max() < 1e-5 and int(done[0]) == int(g["done"][t]) and list(info[0]) == list(g["info"][t]), t
    assert flips == ([92] if path.endswith("hostage_fuzz_03.npz") else []), flips


def test_reset_sampling_ranges_and_key_persistence():
    o = ho.HostageOracle(3, 10, 5, 2, 2, n_envs=512, seed=3)
    o.reset()
    s = o.get_state()
    P = s["pos"]
    assert (P[:, :3, 1] >= 0.55).all() and (P[:, :3, 1] <= 0.95).all()            # rescuers :151
    assert (P[:, :3, 0] >= 0.515).all()                                            # ... then clipped by the closed gate in x too (G3)
    assert (P[:, 3:13, 1] <= 0.36).all() and (P[:, :13] >= 0).all() and (P[:, :13] <= 1).all()    # hostages :158
    assert (P[:, 13:] >= 0).all() and (P[:, 13:] <= 1.01).all()                    # criminals already moved once
    assert (s["vel"][:, :13] == 0).all() and (np.abs(s["vel"][:, 13:]) <= 0.01).all() and (s["vel"][:, 13:] >= 0).mean() > 0.95  # :167 (not centred)
    assert (s["key"] >= 0.9).all() and (s["key"] <= 1.0).all() and (s["bomb"] <= 0.25).all()
    assert ((s["flags"] & 6) == 4).all() and (s["flags"] == 4).mean() > 0.95 and (s["t"] == 1).all()                         # reset ends with one zero-action step
    key0 = s["key"].copy()
    o.reset()
    s2 = o.get_state()
    assert np.array_equal(s2["key"], key0) and not np.array_equal(s2["pos"], P)    # G2: the key is sampled once
