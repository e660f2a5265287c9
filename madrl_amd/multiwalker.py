"""Batched MultiWalkerEnv on MI355X -- host-side mirror of the reference class
`madrl_environments.walker.multi_walker.MultiWalkerEnv` (multi_walker.py:250-641).

Same constructor arguments / defaults (:256-258), `agents`, `reward_mech`, `reset()`, `step()`:
    reset()        -> obs float32 [N, W, 32]  (71 with one_hot ids)
    step(actions)  -> obs, rew float32 [N, W], done bool [N], {}      actions float [N, W, 4]
`MultiWalkerEnv(...)` is the N == 1 drop-in with the reference's return types.

The rigid-body dynamics (Box2D in the reference) are restated from scratch in
madrl_amd/csrc/multiwalker_core.hpp; parity of those dynamics with Box2D is UNPINNED.
Everything the reference module itself computes around `world.Step` -- the world reset() builds,
observations, contact flags, rewards, termination -- is checked against recordings of the
unmodified module (tests/test_multiwalker_envlayer.py).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .base import AbstractMAEnv, Agent, SingleEnvDelegate
from .spaces import Box


class BipedalWalker(Agent):
    """multi_walker.py:87-247 (spaces only)."""

    def __init__(self, obs_dim=24 + 4 + 3 + 1):
        self._obs_dim = obs_dim

    @property
    def observation_space(self):
        return Box(low=-np.inf, high=np.inf, shape=(self._obs_dim,))  # :241-243: 24 + 4 + 3 + (MAX_AGENTS if one_hot else 1)

    @property
    def action_space(self):
        return Box(low=-1, high=1, shape=(4,))


class BatchedMultiWalkerEnv(AbstractMAEnv):

    def __init__(self, n_walkers=2, position_noise=1e-3, angle_noise=1e-3, reward_mech='local', forward_reward=1.0,
                 fall_reward=-100.0, drop_reward=-100.0, terminate_on_fall=True, one_hot=False,
                 n_envs=1, device="cuda:0", seed=0, env_id_base=0, max_steps=0, auto_reset=False, max_blocks=0,
                 continuous_physics=True, box2d_polygon_revision=0):
        self._ctor = dict(locals())
        self._ctor.pop("self"); self._ctor.pop("__class__", None)
        self.n_walkers, self.position_noise, self.angle_noise = n_walkers, position_noise, angle_noise
        self._reward_mech, self.forward_reward, self.fall_reward = reward_mech, forward_reward, fall_reward
        self.drop_reward, self.terminate_on_fall, self.one_hot = drop_reward, terminate_on_fall, one_hot
        self.n_envs, self.device = int(n_envs), torch.device(device)
        self._seed_value, self.env_id_base = int(seed), int(env_id_base)
        self.max_steps, self.auto_reset = int(max_steps), bool(auto_reset)   # (max_blocks: accepted for old call sites, unused -- a launch is one wavefront per group of envs)
        self.continuous_physics = bool(continuous_physics)  # b2World.continuousPhysics (Box2D default True; the reference never changes it)
        # which b2CollidePolygons the hull / package contacts go through: 0 = Box2D 2.3.0 (default), 1 = later 2.3.x revisions (include/madrl_hip.h)
        self.box2d_polygon_revision = int(box2d_polygon_revision)
        self._handle = None
        self.setup()

    def _config(self):
        c = _lib.MultiWalkerConfig()
        c.struct_size = C.sizeof(_lib.MultiWalkerConfig)
        c.n_walkers, c.reward_global = int(self.n_walkers), int(self._reward_mech != "local")
        c.terminate_on_fall, c.one_hot = int(bool(self.terminate_on_fall)), int(bool(self.one_hot))
        c.max_steps, c.auto_reset = self.max_steps, int(self.auto_reset)
        c.discrete_only = 0 if getattr(self, "continuous_physics", True) else 1
        c.polygon_revision = 1 if getattr(self, "box2d_polygon_revision", 0) else 0
        c.position_noise, c.angle_noise = float(self.position_noise), float(self.angle_noise)
        c.forward_reward, c.fall_reward, c.drop_reward = float(self.forward_reward), float(self.fall_reward), float(self.drop_reward)
        c.seed, c.env_id_base = self._seed_value, self.env_id_base
        return c

    def setup(self):
        """multi_walker.py:276-303 (called by set_param_values for curriculum updates)."""
        L = _lib.lib()
        if self.device.type != "cuda":
            raise _lib.MadrlError("BatchedMultiWalkerEnv needs a ROCm device (got %s); there is no CPU path" % self.device)
        cfg = self._config()
        nbytes = C.c_uint64()
        _lib.check(L.madrl_multiwalker_state_bytes(C.byref(cfg), self.n_envs, C.byref(nbytes)))
        N, W, dev = self.n_envs, int(self.n_walkers), self.device
        dim = C.c_int32()
        _lib.check(L.madrl_multiwalker_obs_dim(C.byref(cfg), C.byref(dim)))
        self.obs_dim = dim.value
        if getattr(self, "_shape_key", None) != (N, W, nbytes.value, dim.value):
            self._state = torch.zeros(nbytes.value, dtype=torch.uint8, device=dev)
            self._obs = torch.zeros((N, W, dim.value), dtype=torch.float32, device=dev)
            self._rew = torch.zeros((N, W), dtype=torch.float32, device=dev)
            self._done = torch.zeros(N, dtype=torch.uint8, device=dev)
            self._shape_key = (N, W, nbytes.value, dim.value)
        self._destroy()
        h = C.c_void_p()
        dev_index = dev.index if dev.index is not None else torch.cuda.current_device()
        _lib.check(L.madrl_multiwalker_create(C.byref(cfg), N, dev_index, _lib.ptr(self._state), C.byref(h)))
        self._handle = h
        if getattr(self, "_mode", (0, 1)) != (0, 1):
            _lib.check(L.madrl_multiwalker_set_mode(h, *self._mode))
        nb, nt = C.c_int32(), C.c_int32()
        _lib.check(L.madrl_multiwalker_dims(h, C.byref(nb), C.byref(nt)))
        self.n_bodies, self.n_terrain = nb.value, nt.value
        stride, wb = C.c_int32(), C.c_int32()
        _lib.check(L.madrl_multiwalker_record_bytes(h, C.byref(stride), C.byref(wb)))
        self.record_stride, self.world_bytes = stride.value, wb.value   # block per env (world record + step scratch), world record alone
        capw, lanes = C.c_int32(), C.c_int32()
        _lib.check(L.madrl_multiwalker_lanes(h, C.byref(capw), C.byref(lanes)))
        self.capacity_walkers, self.lanes_per_env = capw.value, lanes.value   # the capacity class of the kernels this walker count runs on
        self.walkers = [BipedalWalker(self.obs_dim) for _ in range(W)]
        self.package_scale = W / 1.75
        self.package_length = 240 / 30.0 * self.package_scale
        self.total_agents = W

    def set_mode(self, fused=False, use_spares=True):
        """How a step is issued (same results either way): `fused` = one launch per b2World::Step instead of three; `use_spares` =
        auto-reset through the episodes prepared ahead of time (False: every auto-reset takes the second pass)."""
        self._mode = (int(bool(fused)), int(bool(use_spares)))
        _lib.check(_lib.lib().madrl_multiwalker_set_mode(self._handle, *self._mode))

    def _destroy(self):
        if getattr(self, "_handle", None):
            _lib.lib().madrl_multiwalker_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    @property
    def agents(self):
        return self.walkers

    @property
    def reward_mech(self):
        return self._reward_mech

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
        _lib.check(_lib.lib().madrl_multiwalker_reset(self._handle, _lib.ptr(mask), _lib.ptr(self._obs),
                                                      _lib.current_stream(self.device)))
        return self._obs

    def step(self, actions, rew_out=None, done_out=None):
        """multi_walker.py:359-428.  done = bit 0 of the kernel's done byte (game over / package dropped, :404-424); bit 1 =
        the max_steps time limit.  With auto_reset either of them starts a new episode, so info carries the raw bits like
        BatchedPursuitEvade does (the rollout collector and the wrappers cut episodes on info['done_bits']).  Bit 7 (info['overflow']) =
        the episode ran out of a capacity -- a contact did not fit its cache or the step's manifold pool and was ignored; sticky until
        the env's next reset, starts no episode by itself.  The pools are sized for walking and falling walkers (the largest
        manifold count seen in random and gait rollouts is about two thirds of them); every walker of eight lying in a heap with
        terminate_on_fall off exceeds them after a few hundred steps.
        rew_out float32 [N, W] / done_out uint8 [N]: optional destinations the kernel writes instead of the env's buffers."""
        N, W = self.n_envs, int(self.n_walkers)
        a = torch.as_tensor(actions, device=self.device)
        if a.numel() != N * W * 4:
            raise AssertionError("actions have %d elements, expected %d" % (a.numel(), N * W * 4))  # :360-361
        a = a.reshape(N, W, 4).to(torch.float32).contiguous()
        rew = self._rew if rew_out is None else rew_out
        dn = self._done if done_out is None else done_out
        assert rew.dtype == torch.float32 and rew.numel() == N * W and dn.dtype == torch.uint8 and dn.numel() == N
        _lib.check(_lib.lib().madrl_multiwalker_step(self._handle, _lib.ptr(a), _lib.ptr(self._obs), _lib.ptr(rew),
                                                     _lib.ptr(dn), _lib.current_stream(self.device)))
        return self._obs, rew, (dn & 1).bool(), {"done_bits": dn, "truncated": (dn & 2).bool(), "overflow": (dn & 128).bool()}

    def bodies(self):
        N, W, dev = self.n_envs, int(self.n_walkers), self.device
        b = torch.zeros((N, self.n_bodies, 6), dtype=torch.float32, device=dev)
        f = torch.zeros((N, 1 + 3 * W), dtype=torch.uint8, device=dev)
        t = torch.zeros((N, self.n_terrain), dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().madrl_multiwalker_get_bodies(self._handle, _lib.ptr(b), _lib.ptr(f), _lib.ptr(t),
                                                           _lib.current_stream(self.device)))
        return b, f, t

    def get_state(self):
        """Unpacked world state (include/madrl_hip.h madrl_multiwalker_get_state): dict of tensors bodies [N, NB, 6], joints
        [N, 4W, 6], aux [N, NB, 6] (fat AABB, sleep time, awake), flags [N, 2 + 3W], terrain [N, NT]."""
        N, W, dev = self.n_envs, int(self.n_walkers), self.device
        st = dict(bodies=torch.zeros((N, self.n_bodies, 6), dtype=torch.float32, device=dev),
                  joints=torch.zeros((N, 4 * W, 6), dtype=torch.float32, device=dev),
                  aux=torch.zeros((N, self.n_bodies, 6), dtype=torch.float32, device=dev),
                  flags=torch.zeros((N, 2 + 3 * W), dtype=torch.uint8, device=dev),
                  terrain=torch.zeros((N, self.n_terrain), dtype=torch.float32, device=dev))
        _lib.check(_lib.lib().madrl_multiwalker_get_state(self._handle, *[_lib.ptr(st[k]) for k in ("bodies", "joints", "aux", "flags", "terrain")],
                                                          _lib.current_stream(self.device)))
        return st

    def set_state(self, bodies=None, joints=None):
        """Overwrite body poses / velocities [N, NB, 6] and, optionally, the joints' accumulated impulses [N, 4W, 6]."""
        conv = lambda a, shape: None if a is None else torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a, device=self.device).reshape(shape).to(torch.float32).contiguous()
        b, j = conv(bodies, (self.n_envs, self.n_bodies, 6)), conv(joints, (self.n_envs, 4 * int(self.n_walkers), 6))
        self._keepalive = (b, j)
        _lib.check(_lib.lib().madrl_multiwalker_set_state(self._handle, _lib.ptr(b), _lib.ptr(j), _lib.current_stream(self.device)))

    def reset_with(self, mask=None, terrain=None, push=None):
        """reset() with the random draws given (parity hook): terrain float64 [N, NT], push float64 [N, W]."""
        conv = lambda a, dt, shape: None if a is None else torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a, device=self.device).reshape(shape).to(dt).contiguous()
        m, t, p = conv(mask, torch.uint8, (self.n_envs,)), conv(terrain, torch.float64, (self.n_envs, self.n_terrain)), conv(push, torch.float64, (self.n_envs, int(self.n_walkers)))
        self._keepalive = (m, t, p)
        _lib.check(_lib.lib().madrl_multiwalker_reset_with(self._handle, _lib.ptr(m), _lib.ptr(t), _lib.ptr(p), _lib.ptr(self._obs),
                                                           _lib.current_stream(self.device)))
        return self._obs

    # float32 / float64 additions, multiplications, divisions and square roots per env-step under uniform random actions with
    # terminate_on_fall (the bench workload), by n_walkers: every primitive of the step (joint / contact velocity and position solves,
    # sub-step sweeps, GJK and root-finder iterations, narrow phase, lidar, ...) carries a hand count of its arithmetic, and the CPU build
    # of the kernel source counts how often each runs (scripts/mw_stats.cpp, 19 500 env-steps per entry; recounted in round 5 for 1 .. 10 walkers)
    COUNTED_FLOPS = {1: 87999.0, 2: 186156.0, 3: 259551.0, 4: 337449.0, 5: 409865.0, 6: 485901.0, 7: 561309.0, 8: 632568.0, 9: 707722.0, 10: 784104.0}

    def flops_per_env_step(self):
        """(floating-point operations per env-step, how the figure was obtained) for the roofline line of bench.py"""
        return self.COUNTED_FLOPS[int(self.n_walkers)], ("counted: hand counts of the arithmetic of every primitive x how often the CPU build of the "
                                                         "kernel source runs it under the bench's action distribution (scripts/mw_stats.cpp)")

    @property
    def state_buffer(self):
        """raw per-env blocks, uint8 [N, record_stride]: the world record (first world_bytes) and the step scratch (checkpoint hook)"""
        return self._state[:self.n_envs * self.record_stride].view(self.n_envs, self.record_stride)

    def __getstate__(self):
        return dict(self._ctor)

    def __setstate__(self, d):
        self.__init__(**d)


class MultiWalkerEnv(SingleEnvDelegate, AbstractMAEnv):
    """N == 1 drop-in with the reference's return types (multi_walker.py:250)."""

    def __init__(self, *args, **kwargs):
        kwargs.pop("n_envs", None)
        self._env = BatchedMultiWalkerEnv(*args, n_envs=1, **kwargs)
        self.reset()  # the reference constructor ends in setup() -> reset() (:271, :303)

    def _after_unpickle(self):
        self.reset()  # EzPickle re-runs the constructor, which ends in reset() (:271, :303)

    def _after_set_params(self):
        self.reset()  # setup() ends in reset() (:303)

    @property
    def agents(self):
        return self._env.agents

    @property
    def reward_mech(self):
        return self._env.reward_mech

    def seed(self, seed=None):
        return self._env.seed(seed)

    def _obslist(self, obs):
        o = obs[0].detach().cpu().numpy().astype(np.float64)
        return [o[i] for i in range(o.shape[0])]

    def reset(self):
        return self._obslist(self._env.reset())

    def step(self, actions):
        act_vec = np.reshape(np.asarray(actions, dtype=np.float64), (self._env.n_walkers, 4))  # :360
        obs, rew, done, info = self._env.step(act_vec[None])
        r = rew[0].detach().cpu().numpy().astype(np.float64)
        rewards = r if self._env.reward_mech == "local" else [float(r[0])] * self._env.n_walkers  # :426-428
        return self._obslist(obs), rewards, bool(done[0].item()), {}
