"""ctypes wrapper over oracle/pursuit_oracle.c.  TEST INFRASTRUCTURE ONLY (see the
header of pursuit_oracle.c): imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by madrl_amd."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "_build", "libmadrl_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(("_oracle.c", "_oracle.cpp"))]
    srcs.append(os.path.join(os.path.dirname(_HERE), "madrl_amd", "csrc", "multiwalker_core.hpp"))
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.po_create.restype = C.c_void_p
        _LIB.po_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_uint64, C.c_int64]
        _LIB.po_destroy.argtypes = [C.c_void_p]
        _LIB.po_reset.argtypes = [C.c_void_p] * 5
        _LIB.po_step.argtypes = [C.c_void_p] * 7
        _LIB.po_get_state.argtypes = [C.c_void_p] * 8
        _LIB.po_set_state.argtypes = [C.c_void_p] * 8
        _LIB.po_get_local_obs.argtypes = [C.c_void_p] * 2
        _LIB.po_set_local_obs.argtypes = [C.c_void_p] * 2
        _LIB.po_obs_dim_of.argtypes = [C.c_void_p]
        _LIB.po_philox4x32_10.argtypes = [C.c_void_p] * 3
        _LIB.po_set_curriculum.argtypes = [C.c_void_p] * 3
        _LIB.po_set_params.argtypes = [C.c_void_p, C.c_double, C.c_double]
    return _LIB


class PoConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "xs", "ys", "n_pursuers", "n_evaders", "obs_range", "n_catch", "surround", "flatten",
        "include_id", "reward_global", "sample_maps", "n_maps", "max_opponents", "train_pursuit")] + [(n, C.c_double) for n in (
            "catchr", "term_pursuit", "urgency_reward", "layer_norm", "constraint_window")]


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def philox(ctr, key):
    ctr = np.asarray(ctr, np.uint32)
    key = np.asarray(key, np.uint32)
    out = np.zeros(4, np.uint32)
    lib().po_philox4x32_10(_p(ctr), _p(key), _p(out))
    return out


class PursuitOracle(object):
    """Batched CPU PursuitEvade with the reference's semantics (kwargs = the reference's)."""

    def __init__(self, map_pool, n_envs=1, seed=0, env_id_base=0, **kw):
        maps = np.ascontiguousarray(np.stack([np.asarray(m) for m in map_pool]).astype(np.int8))
        self.maps = maps
        n_maps, xs, ys = maps.shape
        random_opp, mo = bool(kw.pop("random_opponents", False)), int(kw.pop("max_opponents", 10))
        max_opp = mo if random_opp else 0   # pursuit_evade.py:177-181 (train_pursuit)
        self.cfg = PoConfig(
            xs=xs, ys=ys, n_pursuers=kw.pop("n_pursuers", 1), n_evaders=kw.pop("n_evaders", 1),
            obs_range=kw.pop("obs_range", 3), n_catch=kw.pop("n_catch", 2),
            surround=int(kw.pop("surround", True)), flatten=int(kw.pop("flatten", True)),
            include_id=int(kw.pop("include_id", True)),
            reward_global=int(kw.pop("reward_mech", "global") == "global"),
            sample_maps=int(kw.pop("sample_maps", False)), n_maps=n_maps,
            max_opponents=max_opp, train_pursuit=int(kw.pop("train_pursuit", True)),
            catchr=kw.pop("catchr", 0.01), term_pursuit=kw.pop("term_pursuit", 5.0),
            urgency_reward=kw.pop("urgency_reward", 0.0), layer_norm=kw.pop("layer_norm", 10),
            constraint_window=kw.pop("constraint_window", 1.0))
        assert not kw, "unsupported kwargs: %r" % (kw,)
        self.N = int(n_envs)
        self.P, self.E = self.cfg.n_pursuers, self.cfg.n_evaders
        self.D = lib().po_obs_dim_of(C.byref(self.cfg))
        self.h = lib().po_create(C.byref(self.cfg), _p(maps), self.N, int(seed), int(env_id_base))
        self.obs = np.zeros((self.N, self.P, self.D), np.float32)
        self.rew = np.zeros((self.N, self.P), np.float64)
        self.done = np.zeros(self.N, np.uint8)
        self.removed = np.zeros(self.N, np.int32)

    def __del__(self):
        if getattr(self, "h", None):
            lib().po_destroy(self.h)
            self.h = None

    def reset(self, mask=None, inj_pos=None, inj_map=None):
        if mask is not None:
            mask = np.ascontiguousarray(mask, np.uint8)
        if inj_pos is not None:
            inj_pos = np.ascontiguousarray(inj_pos, np.int32).reshape(self.N, self.P + self.E, 2)
        if inj_map is not None:
            inj_map = np.ascontiguousarray(inj_map, np.int32).reshape(self.N)
        lib().po_reset(self.h, _p(mask), _p(inj_pos), _p(inj_map), _p(self.obs))
        return self.obs

    def step(self, actions, inj_evader_actions=None):
        actions = np.ascontiguousarray(actions, np.int32).reshape(self.N, self.P)
        if inj_evader_actions is not None:  # evader control: one entry per pursuer (the opponents)
            inj_evader_actions = np.ascontiguousarray(inj_evader_actions, np.int32).reshape(self.N, self.E if self.cfg.train_pursuit else self.P)
        lib().po_step(self.h, _p(actions), _p(inj_evader_actions), _p(self.obs), _p(self.rew),
                      _p(self.done), _p(self.removed))
        return self.obs, self.rew, self.done, self.removed

    def set_curriculum(self, constraint_window=None, catchr=None):
        """per-env float64 [N] arrays (kept alive here), None = the config scalar"""
        self._cw = None if constraint_window is None else np.ascontiguousarray(constraint_window, np.float64).reshape(self.N)
        self._cr = None if catchr is None else np.ascontiguousarray(catchr, np.float64).reshape(self.N)
        lib().po_set_curriculum(self.h, _p(self._cw), _p(self._cr))

    def set_params(self, catchr, constraint_window):
        lib().po_set_params(self.h, float(catchr), float(constraint_window))

    def get_state(self):
        N, P, E = self.N, self.P, self.E
        st = dict(pos_p=np.zeros((N, P, 2), np.int32), pos_e=np.zeros((N, E, 2), np.int32),
                  gone=np.zeros((N, E), np.uint8), term_p=np.zeros((N, P), np.uint8),
                  term_e=np.zeros((N, E), np.uint8), map_id=np.zeros(N, np.int32),
                  tick=np.zeros(N, np.uint32))
        lib().po_get_state(self.h, *[_p(st[k]) for k in ("pos_p", "pos_e", "gone", "term_p", "term_e", "map_id", "tick")])
        return st

    def set_state(self, st):
        arrs = [np.ascontiguousarray(st[k], dt) for k, dt in (
            ("pos_p", np.int32), ("pos_e", np.int32), ("gone", np.uint8), ("term_p", np.uint8),
            ("term_e", np.uint8), ("map_id", np.int32), ("tick", np.uint32))]
        lib().po_set_state(self.h, *[_p(a) for a in arrs])

    def local_obs(self):
        R = self.cfg.obs_range
        out = np.zeros((self.N, self.P, 4, R, R), np.float64)
        lib().po_get_local_obs(self.h, _p(out))
        return out

    def set_local_obs(self, lo):
        """overwrite the persistent float64 local_obs [N, P, 4, R, R] (the effect of a caller's in-place edit of the reference's observation views)"""
        R = self.cfg.obs_range
        lo = np.ascontiguousarray(np.asarray(lo, np.float64).reshape(self.N, self.P, 4, R, R))
        lib().po_set_local_obs(self.h, _p(lo))


def config_from_golden(g):
    """kwargs for PursuitOracle / BatchedPursuitEvade from a tests/golden/pursuit_*.npz."""
    return dict(n_pursuers=int(g["cfg_n_pursuers"]), n_evaders=int(g["cfg_n_evaders"]),
                obs_range=int(g["cfg_obs_range"]), n_catch=int(g["cfg_n_catch"]),
                surround=bool(g["cfg_surround"]), flatten=bool(g["cfg_flatten"]),
                include_id=bool(g["cfg_include_id"]),
                reward_mech="global" if int(g["cfg_reward_global"]) else "local",
                sample_maps=bool(g["cfg_sample_maps"]), catchr=float(g["cfg_catchr"]),
                term_pursuit=float(g["cfg_term_pursuit"]),
                urgency_reward=float(g["cfg_urgency_reward"]), layer_norm=float(g["cfg_layer_norm"]),
                **(dict(train_pursuit=False) if "cfg_train_pursuit" in g and not int(g["cfg_train_pursuit"]) else {}),
                **(dict(random_opponents=True, max_opponents=int(g["cfg_max_opponents"]))
                   if "cfg_random_opponents" in g and int(g["cfg_random_opponents"]) else {}))
