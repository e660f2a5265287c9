"""ctypes wrapper over oracle/multiwalker_ref.c -- the INDEPENDENT Box2D-ordered restatement of MultiWalkerEnv.
TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (no Box2D in this image) -- see the header of multiwalker_ref.c.

Two builds: libm sin / cos like Box2D (`poly=False`) and the product's documented polynomial (`poly=True`)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


class MwrConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_walkers", "reward_global", "terminate_on_fall", "one_hot", "continuous_physics", "polygon_revision")] + [
        (n, C.c_double) for n in ("position_noise", "angle_noise", "forward_reward", "fall_reward", "drop_reward")] + [
            ("seed", C.c_uint64), ("env_id_base", C.c_int64)]


def lib(poly=False):
    if poly not in _LIBS:
        so = os.path.join(_HERE, os.environ.get("MADRL_ORACLE_BUILD", "_build"), "libmadrl_mwref_poly.so" if poly else "libmadrl_mwref.so")
        src = os.path.join(_HERE, "multiwalker_ref.c")
        if not os.path.exists(so) or os.path.getmtime(src) > os.path.getmtime(so):
            subprocess.check_call(["make", "-s", "-C", _HERE] + (["asan"] if os.environ.get("MADRL_ORACLE_BUILD") == "_build_asan" else []))
        L = C.CDLL(so)
        L.mwr_create.restype = C.c_void_p
        L.mwr_create.argtypes = [C.c_void_p, C.c_int64]
        for name, n in (("mwr_destroy", 1), ("mwr_reset", 5), ("mwr_step", 5), ("mwr_get_bodies", 2), ("mwr_set_bodies", 2), ("mwr_get_joints", 2),
                        ("mwr_get_aux", 2), ("mwr_get_flags", 2), ("mwr_get_terrain", 2), ("mwr_dims", 3), ("mwr_obs_dim", 1), ("mwr_model", 2),
                        ("mwr_get_stats", 3)):
            getattr(L, name).argtypes = [C.c_void_p] * n
        L.mwr_get_contacts.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int]
        L.mwr_helloworld.argtypes = [C.c_void_p, C.c_int]
        _LIBS[poly] = L
    return _LIBS[poly]


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def helloworld(steps=60, poly=False):
    out = np.zeros((steps, 3), np.float32)
    lib(poly).mwr_helloworld(_p(out), steps)
    return out


class MultiWalkerRef(object):
    """Batched CPU MultiWalkerEnv; observations and rewards float64 like the reference's Python side."""

    def __init__(self, n_walkers=2, position_noise=1e-3, angle_noise=1e-3, reward_mech="local", forward_reward=1.0, fall_reward=-100.0,
                 drop_reward=-100.0, terminate_on_fall=True, one_hot=False, n_envs=1, seed=0, env_id_base=0, continuous_physics=True, poly=False,
                 polygon_revision=0):
        self.L = lib(poly)
        self.cfg = MwrConfig(n_walkers=n_walkers, reward_global=int(reward_mech != "local"), terminate_on_fall=int(terminate_on_fall),
                             one_hot=int(one_hot), continuous_physics=int(continuous_physics), polygon_revision=int(polygon_revision), position_noise=position_noise,
                             angle_noise=angle_noise, forward_reward=forward_reward, fall_reward=fall_reward, drop_reward=drop_reward,
                             seed=int(seed), env_id_base=int(env_id_base))
        self.N, self.W = int(n_envs), int(n_walkers)
        self.h = self.L.mwr_create(C.byref(self.cfg), self.N)
        assert self.h, "unsupported n_walkers"
        nb, nt = C.c_int32(), C.c_int32()
        self.L.mwr_dims(self.h, C.byref(nb), C.byref(nt))
        self.NB, self.NT, self.D = nb.value, nt.value, self.L.mwr_obs_dim(C.byref(self.cfg))
        self.obs = np.zeros((self.N, self.W, self.D), np.float64)
        self.rew = np.zeros((self.N, self.W), np.float64)
        self.done = np.zeros(self.N, np.uint8)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.mwr_destroy(self.h)
            self.h = None

    def reset(self, mask=None, terrain=None, push=None):
        conv = lambda a, dt, shape: None if a is None else np.ascontiguousarray(np.asarray(a, dt).reshape(shape))
        mask = conv(mask, np.uint8, (self.N,))
        self._keep = (conv(terrain, np.float64, (self.N, self.NT)), conv(push, np.float64, (self.N, self.W)))
        self.L.mwr_reset(self.h, _p(mask), _p(self._keep[0]), _p(self._keep[1]), _p(self.obs))
        return self.obs

    def step(self, actions):
        a = np.ascontiguousarray(np.asarray(actions, np.float32).reshape(self.N, self.W, 4))
        self.L.mwr_step(self.h, _p(a), _p(self.obs), _p(self.rew), _p(self.done))
        return self.obs, self.rew, self.done

    def bodies(self):
        out = np.zeros((self.N, self.NB, 6), np.float32)
        self.L.mwr_get_bodies(self.h, _p(out))
        return out

    def set_bodies(self, b):
        b = np.ascontiguousarray(np.asarray(b, np.float32).reshape(self.N, self.NB, 6))
        self.L.mwr_set_bodies(self.h, _p(b))

    def joints(self):
        out = np.zeros((self.N, 4 * self.W, 6), np.float32)
        self.L.mwr_get_joints(self.h, _p(out))
        return out

    def aux(self):
        out = np.zeros((self.N, self.NB, 6), np.float32)
        self.L.mwr_get_aux(self.h, _p(out))
        return out

    def flags(self):
        out = np.zeros((self.N, 1 + 3 * self.W), np.uint8)
        self.L.mwr_get_flags(self.h, _p(out))
        return out

    def terrain(self):
        out = np.zeros((self.N, self.NT), np.float32)
        self.L.mwr_get_terrain(self.h, _p(out))
        return out

    def contacts(self, n, max_contacts=256):
        ints, flts = np.zeros((max_contacts, 8), np.int32), np.zeros((max_contacts, 4), np.float32)
        k = self.L.mwr_get_contacts(self.h, int(n), _p(ints), _p(flts), max_contacts)
        return ints[:k], flts[:k]

    def model(self):
        out = np.zeros((4, 4), np.float32)
        self.L.mwr_model(self.h, _p(out))
        return out

    def stats(self):
        a, b = C.c_int64(), C.c_int64()
        self.L.mwr_get_stats(self.h, C.byref(a), C.byref(b))
        return dict(toi_events=a.value, contacts_created=b.value)
