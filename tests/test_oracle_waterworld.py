"""CPU tests (-m "not gpu"): the Waterworld C oracle against golden vectors produced by the
unmodified reference (oracle/make_golden_waterworld.py), teacher-forced step by step."""
import glob
import os

import numpy as np
import pytest

from oracle import waterworld as ww

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "waterworld_*.npz")))
gid = lambda p: os.path.basename(p)[:-4]


@pytest.mark.parametrize("path", FILES, ids=gid)
@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-12), (np.float32, 1e-5)], ids=["f64", "f32"])
def test_oracle_matches_reference_golden(path, dtype, tol):
    g = np.load(path)
    o = ww.WaterworldOracle(n_envs=1, dtype=dtype, sensors=g["sensors"], **ww.kwargs_from_golden(g))
    assert o.D == g["obs"].shape[-1]
    for t in range(len(g["pre_t"])):
        o.set_state(pos=g["pre_pos"][t][None], vel=g["pre_vel"][t][None], obst=g["obst"][t][None], t=np.array([g["pre_t"][t]]))
        obs, rew, done, info = o.step(g["act"][t][None], resp=g["resp"][t][None])
        st = o.get_state()
        assert np.abs(st["pos"][0] - g["post_pos"][t]).max() <= tol, "pos step %d" % t
        assert np.abs(st["vel"][0] - g["post_vel"][t]).max() <= tol, "vel step %d" % t
        assert np.abs(obs[0] - g["obs"][t]).max() <= tol, "obs step %d" % t
        assert int(st["t"][0]) == int(g["post_t"][t])
        if not g["is_reset_step"][t]:
            assert np.abs(rew[0] - g["rew"][t]).max() <= tol, "rew step %d" % t
            assert int(done[0]) == int(g["done"][t])
            assert int(info[0, 0]) == int(g["evc"][t]) and int(info[0, 1]) == int(g["poc"][t])


def test_golden_covers_catches_and_respawns():
    g = np.load([p for p in FILES if "c3_catches" in p][0])
    assert np.nansum(g["evc"]) > 10 and np.nansum(g["poc"]) > 10
    assert (g["resp"][..., 0] > -1).sum() > 50
    g = np.load([p for p in FILES if "coop1_fast" in p][0])
    outside = ((g["post_pos"] < 0) | (g["post_pos"] > 1)).any()
    assert outside  # W6: evaders drift out of the arena


def test_oracle_reset_keeps_particles_off_the_obstacle_and_counts_one_step():
    o = ww.WaterworldOracle(5, 10, n_envs=512, seed=3, dtype=np.float32)
    o.reset()
    st = o.get_state()
    assert (st["t"] == 1).all()  # W11: reset() performs one zero-action step
    d = np.linalg.norm(st["pos"][:, :5] - st["obst"][:, None], axis=-1)
    assert (d > 0.2).all()
    o2 = ww.WaterworldOracle(5, 10, n_envs=512, seed=3, dtype=np.float32)
    o2.reset()
    assert np.array_equal(o2.get_state()["pos"], st["pos"])
