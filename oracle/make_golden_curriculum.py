#!/usr/bin/env python
"""Generate tests/golden/curriculum_pursuit.npz: the attribute trajectory of the UNMODIFIED reference
PursuitEvade.update_curriculum (pursuit_evade.py:264-272) over 48 iterations -- constraint_window growing by
curriculum_constrain_rate and clipped to [0, 1], one pursuer and one evader removed every curriculum_remove_every
iterations while more than 4 pursuers remain, catchr switched off after curriculum_turn_off_shaping -- and what the
pickled env carries (__getstate__ :397-411).  TEST INFRASTRUCTURE ONLY."""
import os
import pickle
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_loader  # noqa: E402


def main():
    R = ref_loader.load()
    cfg = dict(n_evaders=30, n_pursuers=8, obs_range=7, constraint_window=0.2, catchr=0.1, curriculum_remove_every=7,
               curriculum_constrain_rate=0.03, curriculum_turn_off_shaping=33)
    env = R["PursuitEvade"]([R["TwoDMaps"].rectangle_map(16, 16)], **cfg)
    T = 48
    rec = dict(cw=np.zeros(T), n_evaders=np.zeros(T, np.int64), n_pursuers=np.zeros(T, np.int64), catchr=np.zeros(T))
    for itr in range(T):
        env.update_curriculum(itr)
        rec["cw"][itr], rec["n_evaders"][itr], rec["n_pursuers"][itr], rec["catchr"][itr] = (
            env.constraint_window, env.n_evaders, env.n_pursuers, env.catchr)
    st = env.__getstate__()
    out = dict(rec)
    for k, v in cfg.items():
        out["cfg_" + k] = np.float64(v) if isinstance(v, float) else np.int64(v)
    out["pickled_keys"] = np.array(sorted(k for k in st if k in ("constraint_window", "n_evaders", "n_pursuers", "catchr")))
    path = os.path.join(os.environ.get("MADRL_GOLDEN_OUT") or os.path.join(os.path.dirname(HERE), "tests", "golden"), "curriculum_pursuit.npz")
    np.savez_compressed(path, **out)
    print("curriculum: cw %.2f -> %.2f, pursuers %d -> %d, catchr %.2f -> %.2f (%d iterations, %.1f KB)" % (
        cfg["constraint_window"], rec["cw"][-1], cfg["n_pursuers"], rec["n_pursuers"][-1], cfg["catchr"], rec["catchr"][-1], T,
        os.path.getsize(path) / 1024.0))


if __name__ == "__main__":
    main()
