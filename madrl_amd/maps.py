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
