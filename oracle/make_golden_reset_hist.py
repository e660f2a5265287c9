#!/usr/bin/env python
"""Occupancy histogram of the UNMODIFIED reference's reset() (pursuit_evade.py:173-207) with a constraint window and a sampled
map pool -- the distribution the free-running reset kernels must sample from.  TEST INFRASTRUCTURE ONLY; runs in
the build container, writes tests/golden/resetdist_pursuit.npz.

    window start  sx, sy ~ U(0, 1 - cw)  -> cells [int(xs sx), int(xs (sx + cw))) x [int(ys sy), int(ys (sy + cw)))
    every agent   uniform over the free cells of the window (rejection sampling, agent_utils.py:31-47)
    map           map_pool[np.random.randint(len(pool))] per reset (:182-183)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_loader  # noqa: E402

OUT = os.path.join(os.environ.get("MADRL_GOLDEN_OUT") or os.path.join(os.path.dirname(HERE), "tests", "golden"), "resetdist_pursuit.npz")
CFG = dict(n_evaders=30, n_pursuers=8, obs_range=7, constraint_window=0.5, sample_maps=True)
N_RESETS_PER_WORKER = 5000
AGENTS = [0, 7, 8, 37]   # agent indices (pursuers first) whose single positions are histogrammed


def work(seed):
    R = ref_loader.load()
    pool = np.load(os.path.join(ref_loader.REFERENCE_ROOT, "maps", "map_pool16.npy"))
    maps = [np.asarray(m, dtype=np.int32) for m in pool]
    np.random.seed(seed)
    env = R["PursuitEvade"](maps, **CFG)
    hist = np.zeros((len(maps), 16 * 16), np.int64)
    nmap = np.zeros(len(maps), np.int64)
    # agents of one reset share its window, so their cells are correlated; the per-agent histograms below (one position per
    # reset each, pooled over the maps) are sets of INDEPENDENT samples a chi-square test may be run on
    one = np.zeros((len(AGENTS), 16 * 16), np.int64)
    for _ in range(N_RESETS_PER_WORKER):
        env.reset()
        mid = [i for i, m in enumerate(maps) if env.map_matrix is m][0]
        nmap[mid] += 1
        k = 0
        for layer in (env.pursuer_layer, env.evader_layer):
            for i in range(layer.n_agents()):
                x, y = layer.get_position(i)
                hist[mid, int(x) * 16 + int(y)] += 1
                if k in AGENTS:
                    one[AGENTS.index(k), int(x) * 16 + int(y)] += 1
                k += 1
    return hist, nmap, one


def main():
    import multiprocessing as mp
    with mp.get_context("fork").Pool(os.cpu_count()) as pool:
        res = pool.map(work, range(100, 100 + os.cpu_count()))
    hist = sum(r[0] for r in res)
    nmap = sum(r[1] for r in res)
    one = sum(r[2] for r in res)
    pool16 = np.load(os.path.join(ref_loader.REFERENCE_ROOT, "maps", "map_pool16.npy"))
    np.savez_compressed(OUT, hist=hist, nmap=nmap, one=one, agents=np.asarray(AGENTS), maps=np.asarray(pool16, dtype=np.int8), constraint_window=np.float64(CFG["constraint_window"]),
                        n_pursuers=np.int64(CFG["n_pursuers"]), n_evaders=np.int64(CFG["n_evaders"]))
    print("resets", int(nmap.sum()), "agents", int(hist.sum()), "per map", nmap.tolist(), "->", OUT)


if __name__ == "__main__":
    main()
