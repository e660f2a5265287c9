"""Batched mirrors of the reference's env wrappers (madrl_environments/__init__.py:143-389):
`ObservationBuffer`, `StandardizedEnv`, `DiagnosticsWrapper`, over any Batched* env of this
package.  Each env instance has its own wrapper state, as if N wrapped reference envs ran side
by side; the arithmetic runs in epilogue kernels behind the C ABI (madrl_wrap_*), float64
running statistics like the reference.  With an `auto_reset` env the episode boundary is the
step whose done flag is set: the observation of that step already belongs to the new episode."""
import time

import numpy as np
import torch

from . import _lib
from .base import AbstractMAEnv, Agent, SingleEnvDelegate
from .spaces import Box


def _stream(env):
    return _lib.current_stream(env.device)


class WrappedAgent(Agent):
    """madrl_environments/__init__.py:122-140"""

    def __init__(self, agent, new_observation_space):
        self._unwrapped = agent
        self._new_observation_space = new_observation_space

    @property
    def observation_space(self):
        return self._new_observation_space

    @property
    def action_space(self):
        return self._unwrapped.action_space


class _SingleAsBatched(object):
    """Adapter under a wrapper that was handed an N == 1 drop-in env (reference return types: lists of float64
    arrays, python bool): presents it as a one-env batched env, so the same epilogue kernels serve both.  The drop-in's
    own reset()/step() still run (action parsing, evader_controller, scripted positions)."""

    def __init__(self, env):
        self.single, self.device, self.n_envs, self.auto_reset = env, env.device, 1, False

    def __getattr__(self, name):
        return getattr(self.__dict__["single"], name)

    def _obs(self, obs):
        return torch.as_tensor(np.stack([np.asarray(o) for o in obs])[None], dtype=torch.float32, device=self.device)

    def reset(self, **kw):
        return self._obs(self.single.reset(**kw))

    def step(self, action, **kw):
        obs, rew, done, info = self.single.step(action, **kw)
        r = torch.as_tensor(np.asarray(rew, dtype=np.float64)[None], dtype=torch.float32, device=self.device)
        return self._obs(obs), r, torch.as_tensor([bool(done)], device=self.device), info


def _rows(t):
    o = t[0].detach().cpu().numpy().astype(np.float64)
    return [o[i] for i in range(o.shape[0])]


class _Wrapper(AbstractMAEnv):
    """Every wrapper computes on the batched (tensor) side: `_reset_b` / `_step_b`.  Around an N == 1 drop-in env
    (or a wrapper of one) the public reset()/step() hand back the reference's types instead -- lists of per-agent
    float64 arrays, python bool -- so `StandardizedEnv(PursuitEvade(...))` reads like the reference's."""

    def __init__(self, env):
        self._single = isinstance(env, SingleEnvDelegate) or bool(getattr(env, "_single", False))
        self._unwrapped = _SingleAsBatched(env) if isinstance(env, SingleEnvDelegate) else env
        self.device = self._unwrapped.device
        self.n_envs = self._unwrapped.n_envs

    def _inner_reset(self, **kw):
        u = self._unwrapped
        return u._reset_b(**kw) if isinstance(u, _Wrapper) else u.reset(**kw)

    def _inner_step(self, *args, **kw):
        u = self._unwrapped
        return u._step_b(*args, **kw) if isinstance(u, _Wrapper) else u.step(*args, **kw)

    def reset(self, **kw):
        obs = self._reset_b(**kw)
        return _rows(obs) if self._single else obs

    def step(self, *args, **kw):
        obs, rew, done, info = self._step_b(*args, **kw)
        if not self._single:
            return obs, rew, done, info
        r = rew[0].detach().cpu().numpy().astype(np.float64)
        return _rows(obs), [float(x) for x in r], bool(done[0].item()), self._single_info(info)

    def _single_info(self, info):
        return info

    @property
    def unwrapped(self):
        u = self._unwrapped
        return u.single if isinstance(u, _SingleAsBatched) else u

    @property
    def agents(self):
        return self._unwrapped.agents

    @property
    def reward_mech(self):
        return self._unwrapped.reward_mech

    def seed(self, seed=None):
        return self._unwrapped.seed(seed)

    def set_param_values(self, lut):
        self._unwrapped.set_param_values(lut)

    def _done_u8(self, done, info):
        if isinstance(info, dict) and "done_bits" in info:
            return info["done_bits"]
        return done.to(torch.uint8).contiguous()


class ObservationBuffer(_Wrapper):
    """:143-201 -- keeps the last `buffer_size` observations of every agent, newest last:
    obs float32 [N, A, D, buffer_size].  (The reference's `agents` property has a typo, `ent`, at
    :158; the intended behaviour -- the space grows a trailing axis -- is implemented.)"""

    def __init__(self, env, buffer_size):
        super().__init__(env)
        self._buffer_size = int(buffer_size)
        assert all(len(a.observation_space.shape) == 1 for a in env.agents)  # :148
        self._buf = None

    @property
    def agents(self):
        out = []
        for a in self._unwrapped.agents:
            sp = a.observation_space
            out.append(WrappedAgent(a, Box(low=float(np.min(sp.low)), high=float(np.max(sp.high)),
                                           shape=tuple(sp.shape) + (self._buffer_size,))))
        return out

    def _push(self, obs, reset_mask, active_mask=None):
        obs = obs.contiguous()
        if self._buf is None:
            self._buf = torch.zeros(tuple(obs.shape) + (self._buffer_size,), dtype=torch.float32, device=obs.device)  # :150
        n = obs.numel()
        _lib.check(_lib.lib().madrl_wrap_obsbuffer(_lib.ptr(obs), _lib.ptr(self._buf), n, n // self.n_envs, self._buffer_size,
                                                   _lib.ptr(reset_mask), _lib.ptr(active_mask), _stream(self)))
        return self._buf

    def _reset_b(self, mask=None):
        if mask is None:
            return self._push(self._inner_reset(), torch.ones(self.n_envs, dtype=torch.uint8, device=self.device))
        m = torch.as_tensor(mask, device=self.device).reshape(self.n_envs).to(torch.uint8).contiguous()
        first = self._buf is None
        obs = self._inner_reset(mask=m)
        if first:  # nothing to keep yet: every env starts from its current observation
            return self._push(obs, torch.ones_like(m))
        return self._push(obs, m, active_mask=m)  # envs outside the mask keep their history untouched

    def _step_b(self, action, **kw):
        obs, rew, done, info = self._inner_step(action, **kw)
        reset_mask = self._done_u8(done, info) if getattr(self._unwrapped, "auto_reset", False) else None
        return self._push(obs, reset_mask), rew, done, info


class StandardizedEnv(_Wrapper):
    """:204-311 -- exponential running mean/variance normalisation of observations (per agent,
    per element) and rewards (per agent), plus reward scaling."""

    def __init__(self, env, scale_reward=1., enable_obsnorm=False, enable_rewnorm=False, obs_alpha=0.001, rew_alpha=0.001,
                 eps=1e-8, fused=None):
        """fused: None = fuse into the env's step / reset kernels when the env directly below supports it
        (`bind_standardize`, today the Waterworld engine): the observation row is normalised as it leaves LDS instead of
        being stored raw and read back by an epilogue launch.  False = always the stand-alone epilogue kernels."""
        super().__init__(env)
        self._scale_reward, self._enable_obsnorm, self._enable_rewnorm = scale_reward, enable_obsnorm, enable_rewnorm
        self._obs_alpha, self._rew_alpha, self._eps = obs_alpha, rew_alpha, eps
        self._own = dict(obs_mean=None, obs_var=None, rew_mean=None, rew_var=None)   # epilogue path: this wrapper's own statistics
        self._fused_state = None
        self._obs_out = self._rew_out = None
        self._fused = False
        if fused is not False and not self._single and hasattr(self._unwrapped, "bind_standardize"):
            st = self._unwrapped.bind_standardize(scale_reward=scale_reward, enable_obsnorm=enable_obsnorm, enable_rewnorm=enable_rewnorm,
                                                  obs_alpha=obs_alpha, rew_alpha=rew_alpha, eps=eps)
            self._fused_state = st   # the env re-fills this same dict when a shape change makes it start new statistics
            self._fused = True
        elif fused:
            raise ValueError("fused=True needs an env with bind_standardize() directly below this wrapper")

    def _stat(name):
        def get(self):
            return (self._fused_state if self._fused else self._own)[name]

        def put(self, v):
            (self._fused_state if self._fused else self._own)[name] = v
        return property(get, put)
    _obs_mean, _obs_var, _rew_mean, _rew_var = _stat("obs_mean"), _stat("obs_var"), _stat("rew_mean"), _stat("rew_var")
    del _stat

    def _norm_obs(self, obs):
        if not self._enable_obsnorm:
            return obs
        obs = obs.contiguous()
        if self._obs_mean is None:  # :229-230
            self._obs_mean = torch.zeros(obs.shape, dtype=torch.float64, device=obs.device)
            self._obs_var = torch.ones(obs.shape, dtype=torch.float64, device=obs.device)
            self._obs_out = torch.empty_like(obs)
        n = obs.numel()
        _lib.check(_lib.lib().madrl_wrap_obsnorm(_lib.ptr(obs), _lib.ptr(self._obs_mean), _lib.ptr(self._obs_var), _lib.ptr(self._obs_out),
                                                 n, n // self.n_envs, None, float(self._obs_alpha), float(self._eps), _stream(self)))
        return self._obs_out

    def _norm_rew(self, rew):
        rew = rew.contiguous()
        if self._rew_out is None:  # :231-232
            self._rew_mean = torch.zeros(rew.shape, dtype=torch.float64, device=rew.device)
            self._rew_var = torch.ones(rew.shape, dtype=torch.float64, device=rew.device)
            self._rew_out = torch.empty_like(rew)
        n = rew.numel()
        _lib.check(_lib.lib().madrl_wrap_rewnorm(_lib.ptr(rew), _lib.ptr(self._rew_mean), _lib.ptr(self._rew_var), _lib.ptr(self._rew_out),
                                                 n, n // self.n_envs, None, float(self._rew_alpha), float(self._eps),
                                                 float(self._scale_reward), int(bool(self._enable_rewnorm)), _stream(self)))
        return self._rew_out

    def _reset_b(self, **kw):
        if self._fused:
            return self._inner_reset(**kw)  # already standardised by the env's kernel
        return self._norm_obs(self._inner_reset(**kw))  # :276-281

    def _step_b(self, *args, **kw):
        obs, rew, done, info = self._inner_step(*args, **kw)  # :283-291
        if self._fused:
            return obs, rew, done, info
        return self._norm_obs(obs), self._norm_rew(rew), done, info

    def __str__(self):
        return "Normalized {}".format(self._unwrapped)


class DiagnosticsWrapper(_Wrapper):
    """:314-389 -- per-episode return / discounted return / length, steps-per-second.  step()
    returns the to_log dict of the reference with tensors: values are valid where
    to_log['finished'] is set."""

    def __init__(self, env, discount=0.99, max_traj_len=500, log_interval=501):
        super().__init__(env)
        self._discount, self._max_traj_len, self._log_interval = discount, max_traj_len, log_interval
        N, A, dev = env.n_envs, len(env.agents), env.device
        self._A = A
        f64 = dict(dtype=torch.float64, device=dev)
        self._ep_reward, self._disc_ret, self._disc_pow = torch.zeros((N, A), **f64), torch.zeros(N, **f64), torch.zeros(N, **f64)
        self._ep_len = torch.zeros(N, dtype=torch.int32, device=dev)
        self._out_reward, self._out_disc = torch.zeros((N, A), **f64), torch.zeros(N, **f64)
        self._out_len = torch.zeros(N, dtype=torch.int32, device=dev)
        self._out_fin = torch.zeros(N, dtype=torch.uint8, device=dev)
        self._local_t, self._last_time = 0, time.time()

    def _reset_b(self, **kw):
        obs = self._inner_reset(**kw)  # :328-333
        mask = kw.get("mask")
        if mask is None:
            self._ep_reward.zero_(); self._disc_ret.zero_(); self._ep_len.zero_()
        else:  # partial reset: only the envs that were reset start a new episode
            m = torch.as_tensor(mask, device=self.device).reshape(self.n_envs).bool()
            self._ep_reward[m] = 0; self._disc_ret[m] = 0; self._ep_len[m] = 0
        return obs

    def _single_info(self, to_log):
        """:352-367 -- the reference's to_log dict: scalar entries, present only on the step that ends an episode"""
        out = {k: v for k, v in to_log.items() if k == "diagnostics/fps"}
        if bool(to_log["finished"][0]):
            for a in range(self._A):
                out["global/episode_reward_agent{}".format(a)] = float(to_log["global/episode_reward_agents"][0, a])
            out["global/episode_avg_reward"] = float(to_log["global/episode_avg_reward"][0])
            out["global/episode_disc_return"] = float(to_log["global/episode_disc_return"][0])
            out["global/episode_length"] = int(to_log["global/episode_length"][0])
        return out

    def _step_b(self, *args, **kw):
        obs, rew, done, info = self._inner_step(*args, **kw)
        rew_c = rew.contiguous()
        _lib.check(_lib.lib().madrl_wrap_diagnostics(
            _lib.ptr(rew_c), _lib.ptr(self._done_u8(done, info)), _lib.ptr(self._ep_reward), _lib.ptr(self._ep_len),
            _lib.ptr(self._disc_ret), _lib.ptr(self._disc_pow), self.n_envs, self._A, float(self._discount), int(self._max_traj_len),
            _lib.ptr(self._out_reward), _lib.ptr(self._out_disc), _lib.ptr(self._out_len), _lib.ptr(self._out_fin), _stream(self)))
        to_log = {"finished": self._out_fin.bool(), "global/episode_reward_agents": self._out_reward,
                  "global/episode_avg_reward": self._out_reward.mean(dim=1), "global/episode_disc_return": self._out_disc,
                  "global/episode_length": self._out_len}
        self._local_t += 1
        if self._local_t % self._log_interval == 0:  # :341-347
            now = time.time()
            to_log["diagnostics/fps"] = self._log_interval * self.n_envs / (now - self._last_time)
            self._last_time = now
        return obs, rew, done, to_log
