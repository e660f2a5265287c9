"""Batched ContinuousHostageWorld on MI355X -- host-side mirror of the reference class
`madrl_environments.hostage.ContinuousHostageWorld` (hostage.py:74-430).

* `BatchedContinuousHostageWorld(n_good, n_hostages, n_bad, n_coop_save, n_coop_avoid, ..., n_envs=..., device=...)`: same
  positional / keyword arguments and defaults as the reference constructor (:76-81), same `agents`, `reward_mech`,
  `timestep_limit`, `reset()`, `step()`, `seed()`, `is_terminal`, `is_gate_open`; tensors:
      reset()       -> obs float32 [N, n_good, D]          D = 5K + 6  (K sensors)
      step(action)  -> obs, rew float32 [N, n_good], done bool [N], {'ho_saved','cr_encs': int32 [N]}
* `ContinuousHostageWorld(...)`: N == 1 drop-in with the reference's return types.

Arithmetic is float32 in the HIP kernel (reference: float64; tolerance 1e-5, tests/)."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .base import AbstractMAEnv, Agent, SingleEnvDelegate
from .spaces import Box
from .waterworld import sensor_vectors

_STATE = (("pos", torch.float32), ("vel", torch.float32), ("key", torch.float32), ("bomb", torch.float32), ("saved", torch.int64),
          ("flags", torch.uint8), ("t", torch.int32), ("tick", torch.int32))


class CircAgent(Agent):
    """hostage.py:10-60 (spaces only)."""

    def __init__(self, idx, obs_dim):
        self._idx, self._obs_dim = idx, obs_dim

    @property
    def observation_space(self):
        return Box(low=-np.inf, high=np.inf, shape=(self._obs_dim,))

    @property
    def action_space(self):
        return Box(low=-10, high=10, shape=(2,))


class BatchedContinuousHostageWorld(AbstractMAEnv):

    def __init__(self, n_good, n_hostages, n_bad, n_coop_save, n_coop_avoid, radius=0.015, key_loc=None, bad_speed=0.01, n_sensors=30,
                 sensor_range=0.2, action_scale=0.01, save_reward=5., hit_reward=-1., encounter_reward=0.01, not_saved_reward=-3,
                 bomb_reward=-5., bomb_radius=0.05, key_radius=0.0075, control_penalty=-.1, reward_mech='global', addid=True,
                 n_envs=1, device="cuda:0", seed=0, env_id_base=0, max_steps=0, auto_reset=False, max_blocks=0, **kwargs):
        self._ctor = dict(locals())
        self._ctor.pop("self"); self._ctor.pop("kwargs"); self._ctor.pop("__class__", None)
        self.n_good, self.n_hostages, self.n_bad = n_good, n_hostages, n_bad
        self.n_coop_save, self.n_coop_avoid, self.radius, self.key_loc = n_coop_save, n_coop_avoid, radius, key_loc
        self.key_radius, self.bad_speed, self.n_sensors = key_radius, bad_speed, n_sensors
        self.sensor_range = np.ones(n_good) * sensor_range
        self.action_scale, self.save_reward, self.hit_reward = action_scale, save_reward, hit_reward
        self.encounter_reward, self.not_saved_reward, self.bomb_reward = encounter_reward, not_saved_reward, bomb_reward
        self.bomb_radius, self.control_penalty = bomb_radius, control_penalty
        self._reward_mech, self._addid = reward_mech, addid
        self.n_envs, self.device = int(n_envs), torch.device(device)
        self._seed_value, self.env_id_base = int(seed), int(env_id_base)
        self.max_steps, self.auto_reset, self._max_blocks = int(max_steps), bool(auto_reset), int(max_blocks)
        self._handle = None
        self.setup()

    def _config(self):
        c = _lib.HostageConfig()
        c.struct_size = C.sizeof(_lib.HostageConfig)
        c.n_good, c.n_hostages, c.n_bad = self.n_good, self.n_hostages, self.n_bad
        c.n_coop_save, c.n_coop_avoid, c.n_sensors = self.n_coop_save, self.n_coop_avoid, self.n_sensors
        c.addid, c.reward_global = int(bool(self._addid)), int(self._reward_mech == "global")
        c.key_fixed = int(self.key_loc is not None)
        c.max_steps, c.auto_reset = self.max_steps, int(self.auto_reset)
        c.radius, c.bad_speed = float(self.radius), float(self.bad_speed)
        c.sensor_range, c.action_scale = float(self.sensor_range[0]), float(self.action_scale)
        c.save_reward, c.hit_reward, c.encounter_reward = float(self.save_reward), float(self.hit_reward), float(self.encounter_reward)
        c.not_saved_reward, c.bomb_reward = float(self.not_saved_reward), float(self.bomb_reward)
        c.bomb_radius, c.key_radius, c.control_penalty = float(self.bomb_radius), float(self.key_radius), float(self.control_penalty)
        if self.key_loc is not None:
            k = np.asarray(self.key_loc, np.float64).reshape(2)
            c.key_loc[0], c.key_loc[1] = float(k[0]), float(k[1])
        c.seed, c.env_id_base = self._seed_value, self.env_id_base
        return c

    def setup(self):
        L = _lib.lib()
        if self.device.type != "cuda":
            raise _lib.MadrlError("BatchedContinuousHostageWorld needs a ROCm device (got %s); there is no CPU path" % self.device)
        cfg = self._config()
        dim, nbytes = C.c_int32(), C.c_uint64()
        _lib.check(L.madrl_hostage_obs_dim(C.byref(cfg), C.byref(dim)))
        _lib.check(L.madrl_hostage_state_bytes(C.byref(cfg), self.n_envs, C.byref(nbytes)))
        N, Nr, D, dev = self.n_envs, self.n_good, dim.value, self.device
        self.n_particles = self.n_good + self.n_hostages + self.n_bad
        if getattr(self, "_shape_key", None) != (N, Nr, D, nbytes.value):
            self._state = torch.zeros(nbytes.value, dtype=torch.uint8, device=dev)
            self._obs = torch.zeros((N, Nr, D), dtype=torch.float32, device=dev)
            self._rew = torch.zeros((N, Nr), dtype=torch.float32, device=dev)
            self._done = torch.zeros(N, dtype=torch.uint8, device=dev)
            self._info = torch.zeros((N, 2), dtype=torch.int32, device=dev)
            self._shape_key = (N, Nr, D, nbytes.value)
        self.obs_dim = D
        self._destroy()
        h = C.c_void_p()
        self._sensors = sensor_vectors(self.n_sensors)
        dev_index = dev.index if dev.index is not None else torch.cuda.current_device()
        _lib.check(L.madrl_hostage_create(C.byref(cfg), self._sensors.ctypes.data_as(C.c_void_p), N, dev_index, _lib.ptr(self._state), C.byref(h)))
        self._handle = h
        if self._max_blocks:
            _lib.check(L.madrl_hostage_set_launch(h, self._max_blocks))
        self._rescuers = [CircAgent(i + 1, D) for i in range(Nr)]

    def set_launch(self, max_blocks=0):
        self._max_blocks = int(max_blocks)
        _lib.check(_lib.lib().madrl_hostage_set_launch(self._handle, self._max_blocks))

    def _destroy(self):
        if getattr(self, "_handle", None):
            _lib.lib().madrl_hostage_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    # ------------------------------------------------------------------ reference API
    @property
    def reward_mech(self):
        return self._reward_mech

    @property
    def timestep_limit(self):
        return self.max_steps if self.max_steps > 0 else 1000  # hostage.py:118-120

    @property
    def agents(self):
        return self._rescuers

    def seed(self, seed=None):
        if seed is None:
            seed = int(np.random.randint(2**31 - 1))
        self._seed_value = int(seed)
        self.setup()
        return [self._seed_value]

    def reset(self, mask=None):
        if mask is not None:
            mask = torch.as_tensor(mask, device=self.device).reshape(self.n_envs).to(torch.uint8).contiguous()
        _lib.check(_lib.lib().madrl_hostage_reset(self._handle, _lib.ptr(mask), _lib.ptr(self._obs), _lib.current_stream(self.device)))
        return self._obs

    def step(self, action, respawn=None):
        """hostage.py:228-430.  action: float [N, n_good, 2] (or anything that reshapes to it, :229-230).
        respawn: optional float [N, n_bad, 4] injected respawn uniforms (parity hook)."""
        N, Nr = self.n_envs, self.n_good
        if not self._conforming(action):
            a = torch.as_tensor(action, device=self.device)
            if a.numel() != N * Nr * 2:
                raise AssertionError("action has %d elements, expected %d" % (a.numel(), N * Nr * 2))  # :234
            action = a.reshape(N, Nr, 2).to(torch.float32).contiguous()
        r = None
        if respawn is not None:
            r = torch.as_tensor(respawn, device=self.device).reshape(N, self.n_bad, 4).to(torch.float32).contiguous()
        return self._launch_step(action, r, _lib.current_stream(self.device))

    def _conforming(self, a):
        return (type(a) is torch.Tensor and a.dtype is torch.float32 and a.device == self.device and a.is_contiguous()
                and a.numel() == self.n_envs * self.n_good * 2)

    def step_on_stream(self, action, stream):
        """step() launched on `stream` without making it the current stream (madrl_amd/waterworld.py step_on_stream); None = needs a conversion"""
        if not self._conforming(action):
            return None
        return self._launch_step(action, None, C.c_void_p(stream.cuda_stream))

    def _launch_step(self, a, r, stream_ptr):
        _lib.check(_lib.lib().madrl_hostage_step(self._handle, _lib.ptr(a), _lib.ptr(r), _lib.ptr(self._obs), _lib.ptr(self._rew),
                                                 _lib.ptr(self._done), _lib.ptr(self._info), stream_ptr))
        return self._obs, self._rew, self._done.view(torch.bool), {"ho_saved": self._info[:, 0], "cr_encs": self._info[:, 1], "done_bits": self._done}

    @property
    def is_gate_open(self):
        return (self.get_state()["flags"] & 1).bool()

    @property
    def is_terminal(self):
        s = self.get_state()
        allm = (1 << self.n_hostages) - 1
        return ((s["flags"] & 2) != 0) | ((s["saved"] & allm) == allm) | (s["t"] >= self.timestep_limit)  # :179-182

    def _shapes(self):
        N, NP = self.n_envs, self.n_particles
        return dict(pos=(N, NP, 2), vel=(N, NP, 2), key=(N, 2), bomb=(N, 2), saved=(N,), flags=(N,), t=(N,), tick=(N,))

    def get_state(self):
        sh = self._shapes()
        st = {k: torch.zeros(sh[k], dtype=dt, device=self.device) for k, dt in _STATE}
        _lib.check(_lib.lib().madrl_hostage_get_state(self._handle, *[_lib.ptr(st[k]) for k, _ in _STATE], _lib.current_stream(self.device)))
        return st

    def set_state(self, **kw):
        sh = self._shapes()
        args = []
        for k, dt in _STATE:
            v = kw.get(k)
            if v is not None:
                if not torch.is_tensor(v):
                    v = np.asarray(v)
                    if v.dtype == np.uint64:
                        v = v.astype(np.int64)
                    if v.dtype == np.uint32:
                        v = v.astype(np.int64)
                v = torch.as_tensor(v, device=self.device).reshape(sh[k]).to(dt).contiguous()
            args.append(v)
        self._keepalive = args
        _lib.check(_lib.lib().madrl_hostage_set_state(self._handle, *[_lib.ptr(a) for a in args], _lib.current_stream(self.device)))

    def __getstate__(self):
        return dict(self._ctor)

    def __setstate__(self, d):
        self.__init__(**d)


class ContinuousHostageWorld(SingleEnvDelegate, AbstractMAEnv):
    """N == 1 drop-in with the reference's return types (hostage.py:74)."""

    def __init__(self, *args, **kwargs):
        kwargs.pop("n_envs", None)
        self._env = BatchedContinuousHostageWorld(*args, n_envs=1, **kwargs)

    @property
    def agents(self):
        return self._env.agents

    @property
    def reward_mech(self):
        return self._env.reward_mech

    @property
    def timestep_limit(self):
        return self._env.timestep_limit

    def seed(self, seed=None):
        return self._env.seed(seed)

    def _obslist(self, obs):
        o = obs[0].detach().cpu().numpy().astype(np.float64)
        return [o[i] for i in range(o.shape[0])]

    def reset(self):
        return self._obslist(self._env.reset())

    def step(self, action_Nr2):
        a = np.asarray(action_Nr2, dtype=np.float64).reshape((self._env.n_good, 2))  # :229-230
        obs, rew, done, info = self._env.step(a[None])
        return (self._obslist(obs), rew[0].detach().cpu().numpy().astype(np.float64), bool(done[0].item()),
                dict(ho_saved=int(info["ho_saved"][0].item()), cr_encs=int(info["cr_encs"][0].item())))

    @property
    def is_gate_open(self):
        return bool(self._env.is_gate_open[0].item())

    @property
    def is_terminal(self):
        return bool(self._env.is_terminal[0].item())
