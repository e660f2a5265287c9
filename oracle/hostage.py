"""ctypes wrapper over oracle/hostage_oracle.c.  TEST INFRASTRUCTURE ONLY (see the header of hostage_oracle.c)."""
import ctypes as C

import numpy as np

from . import pursuit as _po
from .waterworld import sensor_vectors, _p


class HwConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_good", "n_hostages", "n_bad", "n_coop_save", "n_coop_avoid", "n_sensors", "addid",
                                         "reward_global", "key_fixed", "max_steps")] + [(n, C.c_double) for n in (
        "radius", "bad_speed", "sensor_range", "action_scale", "save_reward", "hit_reward", "encounter_reward", "not_saved_reward",
        "bomb_reward", "bomb_radius", "key_radius", "control_penalty")] + [("key_loc", C.c_double * 2)]


STATE = (("pos", None), ("vel", None), ("key", None), ("bomb", None), ("saved", np.uint64), ("flags", np.uint8), ("t", np.int32), ("tick", np.uint32))


class HostageOracle(object):
    """Batched CPU ContinuousHostageWorld; dtype float64 = the reference's arithmetic, float32 = the HIP kernel's."""

    def __init__(self, n_good, n_hostages, n_bad, n_coop_save, n_coop_avoid, radius=0.015, key_loc=None, bad_speed=0.01, n_sensors=30,
                 sensor_range=0.2, action_scale=0.01, save_reward=5., hit_reward=-1., encounter_reward=0.01, not_saved_reward=-3,
                 bomb_reward=-5., bomb_radius=0.05, key_radius=0.0075, control_penalty=-.1, reward_mech="global", addid=True,
                 n_envs=1, seed=0, env_id_base=0, max_steps=0, dtype=np.float64, sensors=None):
        L = _po.lib()
        self.dtype = np.dtype(dtype)
        self.pre = "hw64_" if self.dtype == np.float64 else "hw32_"
        c = HwConfig(n_good=n_good, n_hostages=n_hostages, n_bad=n_bad, n_coop_save=n_coop_save, n_coop_avoid=n_coop_avoid,
                     n_sensors=n_sensors, addid=int(addid), reward_global=int(reward_mech == "global"), key_fixed=int(key_loc is not None),
                     max_steps=max_steps, radius=radius, bad_speed=bad_speed, sensor_range=sensor_range, action_scale=action_scale,
                     save_reward=save_reward, hit_reward=hit_reward, encounter_reward=encounter_reward, not_saved_reward=not_saved_reward,
                     bomb_reward=bomb_reward, bomb_radius=bomb_radius, key_radius=key_radius, control_penalty=control_penalty)
        if key_loc is not None:
            k = np.asarray(key_loc, np.float64).reshape(2)
            c.key_loc[0], c.key_loc[1] = float(k[0]), float(k[1])
        self.cfg = c
        self.N, self.Nr, self.Nh, self.Nc = int(n_envs), n_good, n_hostages, n_bad
        self.NP = n_good + n_hostages + n_bad
        fn = lambda name: getattr(L, self.pre + name)
        self._fn = fn
        fn("create").restype = C.c_void_p
        fn("create").argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_uint64, C.c_int64]
        for name, n in (("destroy", 1), ("reset", 3), ("step", 7), ("get_state", 9), ("set_state", 9), ("obs_dim", 1)):
            fn(name).argtypes = [C.c_void_p] * n
        assert fn("real_size")() == self.dtype.itemsize
        self.D = fn("obs_dim")(C.byref(c))
        self.sensors = sensor_vectors(n_sensors) if sensors is None else np.ascontiguousarray(sensors, np.float64)
        self.h = fn("create")(C.byref(c), _p(self.sensors), self.N, int(seed), int(env_id_base))
        self.obs = np.zeros((self.N, self.Nr, self.D), self.dtype)
        self.rew = np.zeros((self.N, self.Nr), self.dtype)
        self.done = np.zeros(self.N, np.uint8)
        self.info = np.zeros((self.N, 2), np.int32)

    def __del__(self):
        if getattr(self, "h", None):
            self._fn("destroy")(self.h)
            self.h = None

    def reset(self, mask=None):
        if mask is not None:
            mask = np.ascontiguousarray(mask, np.uint8)
        self._fn("reset")(self.h, _p(mask), _p(self.obs))
        return self.obs

    def step(self, actions, resp=None):
        a = np.ascontiguousarray(np.asarray(actions, dtype=self.dtype).reshape(self.N, self.Nr, 2))
        if resp is not None:
            resp = np.ascontiguousarray(np.asarray(resp, dtype=self.dtype).reshape(self.N, self.Nc, 4))
        self._fn("step")(self.h, _p(a), _p(resp), _p(self.obs), _p(self.rew), _p(self.done), _p(self.info))
        return self.obs, self.rew, self.done, self.info

    def _shapes(self):
        return dict(pos=(self.N, self.NP, 2), vel=(self.N, self.NP, 2), key=(self.N, 2), bomb=(self.N, 2), saved=(self.N,), flags=(self.N,),
                    t=(self.N,), tick=(self.N,))

    def get_state(self):
        sh = self._shapes()
        st = {k: np.zeros(sh[k], self.dtype if dt is None else dt) for k, dt in STATE}
        self._fn("get_state")(self.h, *[_p(st[k]) for k, _ in STATE])
        return st

    def set_state(self, **kw):
        args = []
        for k, dt in STATE:
            v = kw.get(k)
            args.append(None if v is None else np.ascontiguousarray(v, self.dtype if dt is None else dt))
        self._fn("set_state")(self.h, *[_p(a) for a in args])


def kwargs_from_golden(g):
    kw = {k: int(g["cfg_" + k]) for k in ("n_good", "n_hostages", "n_bad", "n_coop_save", "n_coop_avoid", "n_sensors")}
    kw["addid"] = bool(g["cfg_addid"]); kw["reward_mech"] = "global" if int(g["cfg_reward_global"]) else "local"
    for k in ("radius", "bad_speed", "action_scale", "save_reward", "hit_reward", "encounter_reward", "not_saved_reward", "bomb_reward",
              "bomb_radius", "key_radius", "control_penalty", "sensor_range"):
        kw[k] = float(g["cfg_" + k])
    return kw


def golden_pre_state(g, t):
    """state dict for set_state() from record t of a golden file"""
    saved = np.uint64(sum(int(b) << j for j, b in enumerate(g["pre_saved"][t])))
    flags = np.uint8(int(g["pre_gate"][t]) | (int(g["pre_bombed"][t]) << 1) | 4)
    return dict(pos=g["pre_pos"][t][None], vel=g["pre_vel"][t][None], key=g["key"][t][None], bomb=g["bomb"][t][None], saved=np.array([saved]),
                flags=np.array([flags]), t=np.array([g["pre_t"][t]], np.int32), tick=np.array([t], np.uint32))
