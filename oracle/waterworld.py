"""ctypes wrapper over oracle/waterworld_oracle.c.  TEST INFRASTRUCTURE ONLY (see the header
of waterworld_oracle.c)."""
import ctypes as C

import numpy as np

from . import pursuit as _po


class WwConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n_pursuers", "n_evaders", "n_coop", "n_poison", "n_sensors", "addid", "speed_features",
        "reward_global", "obstacle_fixed", "max_steps")] + [(n, C.c_double) for n in (
            "radius", "obstacle_radius", "ev_speed", "poison_speed", "sensor_range", "action_scale",
            "poison_reward", "food_reward", "encounter_reward", "control_penalty")] + [
                ("obstacle_loc", C.c_double * 2)]


def sensor_vectors(n_sensors):
    """Archea.__init__, waterworld.py:29-31 (float64)"""
    ang = np.linspace(0., 2. * np.pi, n_sensors + 1)[:-1]
    return np.ascontiguousarray(np.c_[np.cos(ang), np.sin(ang)])


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class WaterworldOracle(object):
    """Batched CPU MAWaterWorld.  dtype=np.float64 mirrors the reference's arithmetic,
    dtype=np.float32 the HIP kernel's."""

    def __init__(self, n_pursuers, n_evaders, n_coop=2, n_poison=10, radius=0.015, obstacle_radius=0.2,
                 obstacle_loc=np.array([0.5, 0.5]), ev_speed=0.01, poison_speed=0.01, n_sensors=30,
                 sensor_range=0.2, action_scale=0.01, poison_reward=-1., food_reward=1., encounter_reward=.05,
                 control_penalty=-.5, reward_mech="local", addid=True, speed_features=True, n_envs=1, seed=0,
                 env_id_base=0, max_steps=0, dtype=np.float64, sensors=None):
        L = _po.lib()
        self.dtype = np.dtype(dtype)
        self.pre = "ww64_" if self.dtype == np.float64 else "ww32_"
        c = WwConfig(n_pursuers=n_pursuers, n_evaders=n_evaders, n_coop=n_coop, n_poison=n_poison,
                     n_sensors=n_sensors, addid=int(addid), speed_features=int(speed_features),
                     reward_global=int(reward_mech == "global"), obstacle_fixed=int(obstacle_loc is not None),
                     max_steps=max_steps, radius=radius, obstacle_radius=obstacle_radius, ev_speed=ev_speed,
                     poison_speed=poison_speed, sensor_range=sensor_range, action_scale=action_scale,
                     poison_reward=poison_reward, food_reward=food_reward, encounter_reward=encounter_reward,
                     control_penalty=control_penalty)
        if obstacle_loc is not None:
            c.obstacle_loc[0], c.obstacle_loc[1] = float(obstacle_loc[0]), float(obstacle_loc[1])
        self.cfg = c
        self.N, self.Np, self.NP = int(n_envs), n_pursuers, n_pursuers + n_evaders + n_poison
        fn = lambda name: getattr(L, self.pre + name)
        self._fn = fn
        fn("create").restype = C.c_void_p
        fn("create").argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_uint64, C.c_int64]
        for name, n in (("destroy", 1), ("reset", 3), ("step", 7), ("get_state", 6), ("set_state", 6), ("obs_dim", 1)):
            fn(name).argtypes = [C.c_void_p] * n
        assert fn("real_size")() == self.dtype.itemsize
        self.D = fn("obs_dim")(C.byref(c))
        sens = sensor_vectors(n_sensors) if sensors is None else np.ascontiguousarray(sensors, np.float64)
        self.sensors = sens
        self.h = fn("create")(C.byref(c), _p(sens), self.N, int(seed), int(env_id_base))
        self.obs = np.zeros((self.N, self.Np, self.D), self.dtype)
        self.rew = np.zeros((self.N, self.Np), self.dtype)
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
        a = np.ascontiguousarray(np.asarray(actions, dtype=self.dtype).reshape(self.N, self.Np, 2))
        if resp is not None:
            resp = np.ascontiguousarray(np.asarray(resp, dtype=self.dtype).reshape(self.N, self.NP, 4))
        self._fn("step")(self.h, _p(a), _p(resp), _p(self.obs), _p(self.rew), _p(self.done), _p(self.info))
        return self.obs, self.rew, self.done, self.info

    def get_state(self):
        st = dict(pos=np.zeros((self.N, self.NP, 2), self.dtype), vel=np.zeros((self.N, self.NP, 2), self.dtype),
                  obst=np.zeros((self.N, 2), self.dtype), t=np.zeros(self.N, np.int32), tick=np.zeros(self.N, np.uint32))
        self._fn("get_state")(self.h, *[_p(st[k]) for k in ("pos", "vel", "obst", "t", "tick")])
        return st

    def set_state(self, pos=None, vel=None, obst=None, t=None, tick=None):
        conv = lambda a, dt: None if a is None else np.ascontiguousarray(a, dt)
        args = [conv(pos, self.dtype), conv(vel, self.dtype), conv(obst, self.dtype), conv(t, np.int32), conv(tick, np.uint32)]
        self._fn("set_state")(self.h, *[_p(a) for a in args])


def kwargs_from_golden(g):
    kw = dict(n_pursuers=int(g["cfg_n_pursuers"]), n_evaders=int(g["cfg_n_evaders"]), n_coop=int(g["cfg_n_coop"]),
              n_poison=int(g["cfg_n_poison"]), n_sensors=int(g["cfg_n_sensors"]), addid=bool(g["cfg_addid"]),
              speed_features=bool(g["cfg_speed_features"]),
              reward_mech="global" if int(g["cfg_reward_global"]) else "local")
    for k in ("radius", "obstacle_radius", "ev_speed", "poison_speed", "action_scale", "poison_reward", "food_reward",
              "encounter_reward", "control_penalty", "sensor_range"):
        kw[k] = float(g["cfg_" + k])
    kw["obstacle_loc"] = np.asarray(g["cfg_obstacle_loc"]) if int(g["cfg_obstacle_fixed"]) else None
    return kw
