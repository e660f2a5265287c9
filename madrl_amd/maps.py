"""Host-side map construction (setup code, not on the hot path).

rectangle_map restates utils/TwoDMaps.py:8-22 of the reference: a centred building of -1
cells leaving a fraction xb / yb of the map open on each side."""
import numpy as np


def rectangle_map(xs, ys, xb=0.3, yb=0.2):
    rmap = np.zeros((xs, ys), dtype=np.int32)
    fx = np.arange(xs, dtype=np.float64) / xs
    fy = np.arange(ys, dtype=np.float64) / ys
    inx = (fx > xb) & (fx < (1.0 - xb))
    iny = (fy > yb) & (fy < (1.0 - yb))
    rmap[np.ix_(inx, iny)] = -1
    return rmap


def as_map_pool(map_pool):
    """list / array of (xs,ys) maps with values {0,-1} -> contiguous int8 [n_maps, xs, ys]"""
    if isinstance(map_pool, np.ndarray) and map_pool.ndim == 2:
        map_pool = [map_pool]
    pool = np.ascontiguousarray(np.stack([np.asarray(m) for m in map_pool]).astype(np.int8))
    if pool.ndim != 3:
        raise ValueError("map_pool must be a sequence of 2-D maps")
    if not np.isin(pool, (0, -1)).all():
        raise ValueError("map cells must be 0 (free) or -1 (building)")
    return pool


def resize(scale, maps):
    """utils/TwoDMaps.py:90-94 for an integer scale: `scipy.ndimage.zoom(mat, scale, order=0)` replicates every cell scale x scale
    times (how the survey builds a 32 x 32 pool from the tree's only map file, `resize(2, map_pool16)`)."""
    scale = int(scale)
    return np.stack([np.repeat(np.repeat(np.asarray(m), scale, axis=0), scale, axis=1) for m in maps])


def synthetic_map_pool(n_maps, xs, ys, seed=0, n_rect=(2, 5), side=(0.1, 0.4)):
    """A pool of random maps for benchmarks on boxes without the reference tree (the authors' `map_pool32.npy` / `map_pool128.npy` are
    not in it either): a few axis-aligned buildings per map, about a quarter of the cells built over like `maps/map_pool16.npy`, with
    buildings that may touch row / column 0 (quirk Q4 of need_to_surround).  Synthetic data, not a restatement of a reference file."""
    rng = np.random.RandomState(seed)
    pool = np.zeros((n_maps, xs, ys), dtype=np.int32)
    for m in pool:
        for _ in range(rng.randint(n_rect[0], n_rect[1] + 1)):
            w, h = max(1, int(round(xs * rng.uniform(*side)))), max(1, int(round(ys * rng.uniform(*side))))
            x0, y0 = rng.randint(0, xs - w + 1), rng.randint(0, ys - h + 1)
            m[x0:x0 + w, y0:y0 + h] = -1
    return pool
