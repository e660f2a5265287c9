"""Batched MAWaterWorld on MI355X -- host-side mirror of the reference class
`madrl_environments.pursuit.MAWaterWorld` (waterworld.py:75-436).

* `BatchedMAWaterWorld(n_pursuers, n_evaders, ..., n_envs=..., device=...)`: same positional /
  keyword arguments and defaults as the reference constructor (:77-81), same `agents`,
  `reward_mech`, `timestep_limit`, `reset()`, `step()`, `seed()`, `is_terminal`; tensors:
      reset()       -> obs float32 [N, Np, D]          D = 7K + 3  (K sensors)
      step(action)  -> obs, rew float32 [N, Np], done bool [N], {'evcatches','pocatches': int32 [N]}
* `MAWaterWorld(...)`: N == 1 drop-in with the reference's return types.

Arithmetic is float32 in the HIP kernel (reference: float64; tolerance 1e-5, tests/).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .base import AbstractMAEnv, Agent, SingleEnvDelegate
from .spaces import Box


def sensor_vectors(n_sensors):
    """Archea.__init__, waterworld.py:29-31: unit vectors of the K ray sensors (float64)."""
    angles = np.linspace(0., 2. * np.pi, n_sensors + 1)[:-1]
    return np.ascontiguousarray(np.c_[np.cos(angles), np.sin(angles)])


class Archea(Agent):
    """waterworld.py:10-72 (spaces only)."""

    def __init__(self, idx, obs_dim):
        self._idx = idx
        self._obs_dim = obs_dim

    @property
    def observation_space(self):
        return Box(low=-10, high=10, shape=(self._obs_dim,))

    @property
    def action_space(self):
        return Box(low=-1, high=1, shape=(2,))


class BatchedMAWaterWorld(AbstractMAEnv):

    def __init__(self, n_pursuers, n_evaders, n_coop=2, n_poison=10, radius=0.015, obstacle_radius=0.2,
                 obstacle_loc=np.array([0.5, 0.5]), ev_speed=0.01, poison_speed=0.01, n_sensors=30,
                 sensor_range=0.2, action_scale=0.01, poison_reward=-1., food_reward=1., encounter_reward=.05,
                 control_penalty=-.5, reward_mech='local', addid=True, speed_features=True,
                 n_envs=1, device="cuda:0", seed=0, env_id_base=0, max_steps=0, auto_reset=False, max_blocks=0,
                 **kwargs):
        # like the reference, unknown kwargs are swallowed (waterworld.py:81,:483 passes obs_loc=None)
        self._ctor = dict(locals())
        self._ctor.pop("self"); self._ctor.pop("kwargs"); self._ctor.pop("__class__", None)
        self.n_pursuers, self.n_evaders, self.n_coop, self.n_poison = n_pursuers, n_evaders, n_coop, n_poison
        self.radius, self.obstacle_radius, self.obstacle_loc = radius, obstacle_radius, obstacle_loc
        self.ev_speed, self.poison_speed, self.n_sensors = ev_speed, poison_speed, n_sensors
        self.sensor_range = np.ones(n_pursuers) * sensor_range
        self.action_scale, self.poison_reward, self.food_reward = action_scale, poison_reward, food_reward
        self.control_penalty, self.encounter_reward = control_penalty, encounter_reward
        self.n_obstacles = 1
        self._reward_mech, self._addid, self._speed_features = reward_mech, addid, speed_features
        self.n_envs, self.device = int(n_envs), torch.device(device)
        self._seed_value, self.env_id_base = int(seed), int(env_id_base)
        self.max_steps, self.auto_reset, self._max_blocks = int(max_steps), bool(auto_reset), int(max_blocks)
        self._handle = None
        self.setup()

    def _config(self):
        c = _lib.WaterworldConfig()
        c.struct_size = C.sizeof(_lib.WaterworldConfig)
        c.n_pursuers, c.n_evaders, c.n_coop, c.n_poison = self.n_pursuers, self.n_evaders, self.n_coop, self.n_poison
        c.n_sensors, c.addid, c.speed_features = self.n_sensors, int(bool(self._addid)), int(bool(self._speed_features))
        c.reward_global = int(self._reward_mech == "global")
        c.obstacle_fixed = int(self.obstacle_loc is not None)
        c.max_steps, c.auto_reset = self.max_steps, int(self.auto_reset)
        c.radius, c.obstacle_radius = float(self.radius), float(self.obstacle_radius)
        c.ev_speed, c.poison_speed = float(self.ev_speed), float(self.poison_speed)
        c.sensor_range, c.action_scale = float(self.sensor_range[0]), float(self.action_scale)
        c.poison_reward, c.food_reward = float(self.poison_reward), float(self.food_reward)
        c.encounter_reward, c.control_penalty = float(self.encounter_reward), float(self.control_penalty)
        if self.obstacle_loc is not None:
            c.obstacle_loc[0], c.obstacle_loc[1] = float(self.obstacle_loc[0]), float(self.obstacle_loc[1])
        c.seed, c.env_id_base = self._seed_value, self.env_id_base
        return c

    def setup(self):
        L = _lib.lib()
        if self.device.type != "cuda":
            raise _lib.MadrlError("BatchedMAWaterWorld needs a ROCm device (got %s); there is no CPU path" % self.device)
        cfg = self._config()
        dim, nbytes = C.c_int32(), C.c_uint64()
        _lib.check(L.madrl_waterworld_obs_dim(C.byref(cfg), C.byref(dim)))
        _lib.check(L.madrl_waterworld_state_bytes(C.byref(cfg), self.n_envs, C.byref(nbytes)))
        N, Np, D, dev = self.n_envs, self.n_pursuers, dim.value, self.device
        self.n_particles = self.n_pursuers + self.n_evaders + self.n_poison
        if getattr(self, "_shape_key", None) != (N, Np, D, nbytes.value):
            self._state = torch.zeros(nbytes.value, dtype=torch.uint8, device=dev)
            self._obs = torch.zeros((N, Np, D), dtype=torch.float32, device=dev)
            self._rew = torch.zeros((N, Np), dtype=torch.float32, device=dev)
            self._done = torch.zeros(N, dtype=torch.uint8, device=dev)
            self._info = torch.zeros((N, 2), dtype=torch.int32, device=dev)
            self._shape_key = (N, Np, D, nbytes.value)
        self.obs_dim = D
        self._destroy()
        h = C.c_void_p()
        self._sensors = sensor_vectors(self.n_sensors)
        dev_index = dev.index if dev.index is not None else torch.cuda.current_device()
        _lib.check(L.madrl_waterworld_create(C.byref(cfg), self._sensors.ctypes.data_as(C.c_void_p), N, dev_index,
                                             _lib.ptr(self._state), C.byref(h)))
        self._handle = h
        if self._max_blocks:
            _lib.check(L.madrl_waterworld_set_launch(h, self._max_blocks))
        if N >= 4096:
            self._hint_fast_path(D)
        self._pursuers = [Archea(i + 1, D) for i in range(Np)]
        # A fused StandardizedEnv binding belongs to the handle that was just replaced (seed() and set_param_values() come
        # through here): bind the new handle to the SAME statistics / output tensors, or -- when the shapes changed -- to
        # fresh ones, so that the wrapper keeps receiving standardised rows.
        old, self._std = getattr(self, "_std", None), None
        if old is not None:
            if tuple(old["obs_out"].shape) == (N, Np, D):
                self.bind_standardize(tensors=old, **self._std_kwargs)
            else:   # new shapes: fresh statistics, handed to the wrapper through the SAME dict object it holds
                fresh = self.bind_standardize(tensors=None, **self._std_kwargs)
                old.clear(); old.update(fresh)
                self._std = old

    _hinted = set()

    def _hint_fast_path(self, D):
        """A large batch of a shape that is not in csrc/waterworld_specializations.def runs on the generic instantiation (dynamic LDS layout,
        run-time loop bounds: about half the speed): say once per shape how to give it its own kernel.  Results are identical either way."""
        from . import build as _build
        shape = (int(self.n_pursuers), int(self.n_evaders), int(self.n_poison), int(self.n_sensors), int(D))
        if shape in BatchedMAWaterWorld._hinted or _build.waterworld_is_specialised(*shape):
            return
        import warnings
        BatchedMAWaterWorld._hinted.add(shape)
        warnings.warn("MAWaterWorld with %d pursuers / %d evaders / %d poison / %d sensors (obs_dim %d) runs on the generic kernel; `python -m madrl_amd.build "
                      "--waterworld-shape %d %d %d %d %d` compiles the specialised kernel for this shape (results are identical, a step takes about half the "
                      "time)" % (shape + shape), stacklevel=3)

    def set_launch(self, max_blocks=0):
        self._max_blocks = int(max_blocks)
        _lib.check(_lib.lib().madrl_waterworld_set_launch(self._handle, self._max_blocks))

    def _destroy(self):
        if getattr(self, "_handle", None):
            _lib.lib().madrl_waterworld_destroy(self._handle)
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
        return self.max_steps if self.max_steps > 0 else 1000  # waterworld.py:124-126

    @property
    def agents(self):
        return self._pursuers

    def get_param_values(self):
        return self.__dict__

    def seed(self, seed=None):
        if seed is None:
            seed = int(np.random.randint(2**31 - 1))
        self._seed_value = int(seed)
        self.setup()
        return [self._seed_value]

    def reset(self, mask=None):
        if mask is not None:
            mask = torch.as_tensor(mask, device=self.device).reshape(self.n_envs).to(torch.uint8).contiguous()
        std = getattr(self, "_std", None)
        _lib.check(_lib.lib().madrl_waterworld_reset(self._handle, _lib.ptr(mask), None if std else _lib.ptr(self._obs),
                                                     _lib.current_stream(self.device)))
        return std["obs_out"] if std else self._obs

    # ------------------------------------------------------------------ fused StandardizedEnv (include/madrl_hip.h)
    def bind_standardize(self, scale_reward=1.0, enable_obsnorm=False, enable_rewnorm=False, obs_alpha=0.001, rew_alpha=0.001, eps=1e-8,
                         tensors=None):
        """The kernels normalise observations / rewards on their way out (madrl_waterworld_set_standardize): reset() and
        step() then return the standardised tensors and the raw observation row is not stored.  Returns the dict of
        state tensors (running statistics, outputs) the wrapper owns; `tensors` re-binds an existing dict (setup())."""
        N, Np, D, dev = self.n_envs, self.n_pursuers, self.obs_dim, self.device
        self._std_kwargs = dict(scale_reward=scale_reward, enable_obsnorm=enable_obsnorm, enable_rewnorm=enable_rewnorm,
                                obs_alpha=obs_alpha, rew_alpha=rew_alpha, eps=eps)
        st = tensors if tensors is not None else dict(obs_mean=torch.zeros((N, Np, D), dtype=torch.float64, device=dev), obs_var=torch.ones((N, Np, D), dtype=torch.float64, device=dev),
                  obs_out=torch.zeros((N, Np, D), dtype=torch.float32, device=dev),
                  rew_mean=torch.zeros((N, Np), dtype=torch.float64, device=dev), rew_var=torch.ones((N, Np), dtype=torch.float64, device=dev),
                  rew_out=torch.zeros((N, Np), dtype=torch.float32, device=dev))
        a = _lib.StandardizeArgs()
        a.struct_size = C.sizeof(_lib.StandardizeArgs)
        a.enable_obsnorm, a.enable_rewnorm = int(bool(enable_obsnorm)), int(bool(enable_rewnorm))
        a.obs_alpha, a.rew_alpha, a.eps, a.scale_reward = float(obs_alpha), float(rew_alpha), float(eps), float(scale_reward)
        for k, v in st.items():
            setattr(a, k, v.data_ptr())
        _lib.check(_lib.lib().madrl_waterworld_set_standardize(self._handle, C.byref(a)))
        self._std = st
        return st

    def unbind_standardize(self):
        _lib.check(_lib.lib().madrl_waterworld_set_standardize(self._handle, None))
        self._std = None

    def step(self, action, respawn=None):
        """waterworld.py:220-436.  action: float [N, Np, 2] (or anything that reshapes to it, :221-222).
        respawn: optional float [N, NP, 4] injected respawn outcomes (parity hook)."""
        N, Np = self.n_envs, self.n_pursuers
        if not self._conforming(action):
            a = torch.as_tensor(action, device=self.device)
            if a.numel() != N * Np * 2:
                raise AssertionError("action has %d elements, expected %d" % (a.numel(), N * Np * 2))  # :227
            action = a.reshape(N, Np, 2).to(torch.float32).contiguous()
        r = None
        if respawn is not None:
            r = torch.as_tensor(respawn, device=self.device).reshape(N, self.n_particles, 4).to(torch.float32).contiguous()
        return self._launch_step(action, r, _lib.current_stream(self.device))

    def _conforming(self, a):
        """an action tensor the kernel can read as it is (float32, contiguous, on the device, N * Np * 2 elements): no torch kernel needed"""
        return (type(a) is torch.Tensor and a.dtype is torch.float32 and a.device == self.device and a.is_contiguous()
                and a.numel() == self.n_envs * self.n_pursuers * 2)

    def step_on_stream(self, action, stream):
        """step() launched on `stream` (a torch.cuda.Stream) without making it the current stream -- for callers that drive sub-batches on
        their own streams (madrl_amd/sharded.py): entering a `with torch.cuda.stream(...)` block costs the host more than this launch.
        Returns None when the action needs a conversion kernel (the caller then takes step() under the stream context)."""
        if not self._conforming(action):
            return None
        return self._launch_step(action, None, C.c_void_p(stream.cuda_stream))

    def _launch_step(self, a, r, stream_ptr):
        std = getattr(self, "_std", None)
        _lib.check(_lib.lib().madrl_waterworld_step(self._handle, _lib.ptr(a), _lib.ptr(r), None if std else _lib.ptr(self._obs),
                                                    _lib.ptr(self._rew), _lib.ptr(self._done), _lib.ptr(self._info), stream_ptr))
        # `done` is a bool VIEW of the byte the kernel wrote (0 / 1): no torch kernel runs after the launch
        info = {"evcatches": self._info[:, 0], "pocatches": self._info[:, 1], "done_bits": self._done}
        if std:  # fused StandardizedEnv: standardised observations and scaled / normalised rewards straight from the kernel
            return std["obs_out"], std["rew_out"], self._done.view(torch.bool), info
        return self._obs, self._rew, self._done.view(torch.bool), info

    @property
    def is_terminal(self):
        return self.get_state()["t"] >= self.timestep_limit

    def get_state(self):
        N, NP, dev = self.n_envs, self.n_particles, self.device
        st = dict(pos=torch.zeros((N, NP, 2), dtype=torch.float32, device=dev),
                  vel=torch.zeros((N, NP, 2), dtype=torch.float32, device=dev),
                  obst=torch.zeros((N, 2), dtype=torch.float32, device=dev),
                  t=torch.zeros(N, dtype=torch.int32, device=dev), tick=torch.zeros(N, dtype=torch.int32, device=dev))
        _lib.check(_lib.lib().madrl_waterworld_get_state(self._handle, *[_lib.ptr(st[k]) for k in ("pos", "vel", "obst", "t", "tick")],
                                                         _lib.current_stream(self.device)))
        return st

    def set_state(self, pos=None, vel=None, obst=None, t=None, tick=None):
        N, NP = self.n_envs, self.n_particles
        spec = ((pos, (N, NP, 2), torch.float32), (vel, (N, NP, 2), torch.float32), (obst, (N, 2), torch.float32),
                (t, (N,), torch.int32), (tick, (N,), torch.int32))
        args = []
        for v, shape, dt in spec:
            if v is not None:
                v = torch.as_tensor(np.asarray(v) if not torch.is_tensor(v) else v, device=self.device)
                v = v.reshape(shape).to(dt).contiguous()
            args.append(v)
        self._keepalive = args
        _lib.check(_lib.lib().madrl_waterworld_set_state(self._handle, *[_lib.ptr(a) for a in args],
                                                         _lib.current_stream(self.device)))

    def __getstate__(self):
        return dict(self._ctor)

    def __setstate__(self, d):
        self.__init__(**d)


class MAWaterWorld(SingleEnvDelegate, AbstractMAEnv):
    """N == 1 drop-in with the reference's return types (waterworld.py:75)."""

    def __init__(self, *args, **kwargs):
        kwargs.pop("n_envs", None)
        self._env = BatchedMAWaterWorld(*args, n_envs=1, **kwargs)

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

    def step(self, action_Np2):
        a = np.asarray(action_Np2, dtype=np.float64).reshape((self._env.n_pursuers, 2))  # :221-222
        obs, rew, done, info = self._env.step(a[None])
        return (self._obslist(obs), rew[0].detach().cpu().numpy().astype(np.float64), bool(done[0].item()),
                dict(evcatches=int(info["evcatches"][0].item()), pocatches=int(info["pocatches"][0].item())))

    @property
    def is_terminal(self):
        return bool(self._env.is_terminal[0].item())
