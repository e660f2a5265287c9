#!/usr/bin/env python
"""Golden vectors for the hand-written policies: the UNMODIFIED reference classes
heuristics/pursuit.py:13-56, heuristics/waterworld.py:6-62, heuristics/multi_walker.py:10-86 evaluated on
recorded / synthetic observations.  TEST INFRASTRUCTURE ONLY.

The reference is Python 2 code (`xrange`, integer `/` in `x, y = xs / 2, ys / 2`, heuristics/pursuit.py:23).  It runs here
under Python 3 with the Python 2 meaning restored from the outside: `xrange` is injected into the module globals and the
observation is handed over as an ndarray subclass whose `.shape` entries divide like Python 2 ints."""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_loader  # noqa: E402

OUT = os.path.join(os.environ.get("MADRL_GOLDEN_OUT") or os.path.join(os.path.dirname(HERE), "tests", "golden"))


class Py2Int(int):
    def __truediv__(self, o):
        return Py2Int(int(self) // int(o)) if isinstance(o, int) else int(self) / o


class Py2Array(np.ndarray):
    @property
    def shape(self):
        return tuple(Py2Int(v) for v in np.ndarray.shape.__get__(self))


class FixedSpace(object):
    """action_space whose sample() flags the call (the kernels draw from Philox there)"""

    def sample(self):
        return -1


def main():
    ref_loader.load()
    hp = importlib.import_module("heuristics.pursuit")
    hw = importlib.import_module("heuristics.waterworld")
    hm = importlib.import_module("heuristics.multi_walker")
    hm.xrange = range
    rng = np.random.RandomState(7)
    out = {}
    # ---- pursuit: windows (R, R, 4) channel-last like PursuitEvade(flatten=False) hands them out
    for R in (7, 5, 11):
        wins = []
        for i in range(R):           # every single-evader position
            for j in range(R):
                w = np.zeros((R, R, 4)); w[i, j, 2] = 0.1; wins.append(w)
        for _ in range(600):         # sparse random windows, several evaders (ties included), some empty
            w = np.zeros((R, R, 4))
            k = rng.randint(0, 5)
            for _ in range(k):
                w[rng.randint(R), rng.randint(R), 2] += 0.1
            w[..., 1] = (rng.rand(R, R) < 0.1) * 0.1
            w[..., 0] = (rng.rand(R, R) < 0.1) * 0.1
            wins.append(w)
        wins = np.asarray(wins)
        pol = hp.PursuitHeuristicPolicy(None, FixedSpace())
        acts = np.array([pol.sample_actions(w.view(Py2Array))[0] for w in wins], dtype=np.int32)
        out["pursuit_R%d_obs" % R] = wins.astype(np.float32)
        out["pursuit_R%d_act" % R] = acts      # -1: the reference sampled a random action
    # ---- waterworld: recorded observations of the reference env, one agent row at a time (B = 1)
    g = np.load(os.path.join(OUT, "waterworld_c3_catches.npz"))
    obs = g["obs"].reshape(-1, g["obs"].shape[-1])[:1500]
    extra = obs[:50].copy(); extra[:, :] = 0.0   # all-zero sensors: zero action
    obs = np.concatenate([obs, extra])
    pol = hw.WaterworldHeuristicPolicy(None, None)
    out["waterworld_obs"] = obs.astype(np.float32)
    out["waterworld_act"] = np.concatenate([pol.sample_actions(o.astype(np.float32).astype(np.float64)[None])[0] for o in obs])
    # ---- multiwalker: the policy is a pure function of the 32-vector; synthetic vectors around its thresholds
    obs = rng.uniform(-1.2, 1.2, size=(3000, 32))
    obs[:, 8] = rng.rand(3000) < 0.5; obs[:, 13] = rng.rand(3000) < 0.5
    obs[:500, 2] = rng.uniform(0.2, 0.4, 500)      # around SPEED
    obs[500:1000, 9] = rng.uniform(0.0, 0.2, 500)  # supporting leg behind threshold
    obs[1000:1500, 11] = rng.uniform(0.8, 0.95, 500)
    obs = obs.astype(np.float32)
    pol = hm.MultiWalkerHeuristicPolicy(None, None)
    out["multiwalker_obs"] = obs
    out["multiwalker_act"] = pol.sample_actions(obs.astype(np.float64))[0]
    path = os.path.join(OUT, "heuristics.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KB", {k: v.shape for k, v in out.items()})
    a = out["pursuit_R7_act"]
    print("pursuit R7 action histogram (-1 = random):", {int(k): int((a == k).sum()) for k in np.unique(a)})


if __name__ == "__main__":
    main()
