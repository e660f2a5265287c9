"""Batched PursuitEvade on MI355X -- host-side mirror of the reference class
`madrl_environments.pursuit.PursuitEvade` (pursuit_evade.py:26-541).

Two classes:

* `BatchedPursuitEvade(map_pool, n_envs=..., device=..., **reference_kwargs)` -- the
  vectorised env.  Same constructor kwargs / defaults, same `agents`, `reward_mech`,
  `reset()`, `step()`, `seed()`, `is_terminal`, `set_param_values()`,
  `update_curriculum()`, pickling by constructor arguments; tensors instead of lists:
      reset()        -> obs  float32 [N, P, D]            (D = 3R^2(+1) or (R,R,4))
      step(actions)  -> obs, rew float32 [N, P], done bool [N], {'removed': int32 [N], ...}
* `PursuitEvade(map_pool, **reference_kwargs)` -- N == 1 drop-in with the reference's
  exact return types (list of per-agent ndarrays, ndarray / list rewards, bool, dict), for
  the in-tree callers: rllabwrapper/__init__.py:75-82, heuristics/pursuit.py:71-85,
  madrl_environments/__init__.py:72-109.

All arithmetic runs in the HIP kernels behind the C ABI (include/madrl_hip.h); this file
only owns buffers (torch tensors) and argument checking.  No CPU fallback.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .base import AbstractMAEnv, Agent, SingleEnvDelegate
from .maps import as_map_pool
from .spaces import Box, Discrete

_DEFAULTS = dict(  # pursuit_evade.py:49-148
    sample_maps=False, reward_mech="global", n_evaders=1, n_pursuers=1, obs_range=3,
    flatten=True, layer_norm=10, n_catch=2, random_opponents=False, max_opponents=10,
    catchr=0.01, caughtr=-0.01, term_pursuit=5.0, term_evade=-5.0, urgency_reward=0.0,
    include_id=True, train_pursuit=True, initial_config=None, surround=True,
    constraint_window=1.0, curriculum_remove_every=500, curriculum_constrain_rate=0.0,
    curriculum_turn_off_shaping=np.inf)


class PursuitAgent(Agent):
    """utils/DiscreteAgent.py:11-66 (spaces only; dynamics live in the kernel)."""

    def __init__(self, obs_shape):
        self._obs_shape = tuple(obs_shape)

    @property
    def observation_space(self):
        return Box(low=-np.inf, high=np.inf, shape=self._obs_shape)

    @property
    def action_space(self):
        return Discrete(5)


class BatchedPursuitEvade(AbstractMAEnv):

    def __init__(self, map_pool, n_envs=1, device="cuda:0", seed=0, env_id_base=0, max_steps=0,
                 auto_reset=False, threads=0, max_blocks=0, kernel="auto", **kwargs):
        self._ctor = dict(map_pool=map_pool, n_envs=n_envs, device=str(device), seed=seed,
                          env_id_base=env_id_base, max_steps=max_steps, auto_reset=auto_reset,
                          threads=threads, max_blocks=max_blocks, kernel=kernel, kwargs=dict(kwargs))
        self._kernel = kernel
        kw = dict(_DEFAULTS)
        for k in list(kwargs):
            if k in ("ally_layer", "opponent_layer", "evader_controller", "pursuer_controller"):
                raise ValueError("%s is a Python object hook of the reference; the batched engine "
                                 "injects evader actions through step(..., evader_actions=)" % k)
            if k not in kw:
                raise TypeError("unknown PursuitEvade kwarg %r" % k)
            kw[k] = kwargs[k]
        if kw["random_opponents"] and int(kw["max_opponents"]) < 2:
            raise ValueError("random_opponents draws randint(1, max_opponents): max_opponents must be >= 2")  # :179
        if not kw["train_pursuit"]:  # evader control, pursuit_evade.py:105-112, :204-207, :215-224
            if kw["random_opponents"]:
                raise NotImplementedError("train_pursuit=False with random_opponents (a per-reset number of pursuers, :180-181)")
            if int(kw["n_evaders"]) < int(kw["n_pursuers"]):
                raise ValueError("train_pursuit=False: collect_obs indexes evaders_gone[i] for i < n_pursuers (:418-428); "
                                 "the reference raises IndexError in reset() when n_evaders < n_pursuers")
        for k, v in kw.items():
            if k == "reward_mech":
                self._reward_mech = v
            else:
                setattr(self, k, v)
        self.map_pool = as_map_pool(map_pool)
        self.map_matrix = self.map_pool[0]
        self.xs, self.ys = self.map_matrix.shape
        self.obs_offset = int((self.obs_range - 1) / 2)
        self.n_envs = int(n_envs)
        self.device = torch.device(device)
        self._seed_value = int(seed)
        self.env_id_base = int(env_id_base)
        self.max_steps = int(max_steps)
        self.auto_reset = bool(auto_reset)
        self._threads, self._max_blocks = int(threads), int(max_blocks)
        self._handle = None
        self.setup()

    # ------------------------------------------------------------------ plumbing
    def _config(self):
        c = _lib.PursuitConfig()
        c.struct_size = C.sizeof(_lib.PursuitConfig)
        c.xs, c.ys = self.xs, self.ys
        c.n_pursuers, c.n_evaders = int(self.n_pursuers), int(self.n_evaders)
        c.obs_range, c.n_catch = int(self.obs_range), int(self.n_catch)
        c.surround, c.flatten, c.include_id = int(bool(self.surround)), int(bool(self.flatten)), int(bool(self.include_id))
        c.reward_global = int(self._reward_mech == "global")
        c.sample_maps, c.n_maps = int(bool(self.sample_maps)), int(self.map_pool.shape[0])
        c.max_steps, c.auto_reset = self.max_steps, int(self.auto_reset)
        # :177-181 (train_pursuit): every reset creates randint(1, max_opponents) evaders, at most the n_evaders slots
        c.max_opponents = int(self.max_opponents) if self.random_opponents else 0
        c.control_evaders = int(not self.train_pursuit)
        c.catchr, c.term_pursuit = float(self.catchr), float(self.term_pursuit)
        c.urgency_reward, c.layer_norm = float(self.urgency_reward), float(self.layer_norm)
        c.constraint_window = float(self.constraint_window)
        c.seed, c.env_id_base = self._seed_value, self.env_id_base
        return c

    def setup(self):
        """(Re)build the native handle from the current attributes.  Called by __init__ and by
        set_param_values (madrl_environments/__init__.py:64-67).  Buffers are kept when the
        shapes did not change, so curriculum updates of catchr / constraint_window are cheap."""
        L = _lib.lib()
        if self.device.type != "cuda":
            raise _lib.MadrlError("BatchedPursuitEvade needs a ROCm device (got %s); there is no "
                                  "CPU path" % self.device)
        cfg = self._config()
        dim = C.c_int32()
        _lib.check(L.madrl_pursuit_obs_dim(C.byref(cfg), C.byref(dim)))
        nbytes, rbytes = C.c_uint64(), C.c_int32()
        _lib.check(L.madrl_pursuit_state_bytes(C.byref(cfg), self.n_envs, C.byref(nbytes)))
        _lib.check(L.madrl_pursuit_record_bytes(C.byref(cfg), C.byref(rbytes)))
        self.record_bytes = rbytes.value
        foff = C.c_uint64()
        _lib.check(L.madrl_pursuit_flags_offset(C.byref(cfg), self.n_envs, C.byref(foff)))
        N, P, E, D = self.n_envs, int(self.n_pursuers), int(self.n_evaders), dim.value
        shape_key = (N, P, E, D, nbytes.value)
        if getattr(self, "_shape_key", None) != shape_key:
            dev = self.device
            self._state = torch.zeros(nbytes.value, dtype=torch.uint8, device=dev)
            # IN/OUT observation buffer == the reference's persistent local_obs (Q2)
            self._obs = torch.zeros((N, P, D), dtype=torch.float32, device=dev)
            self._rew = torch.zeros((N, P), dtype=torch.float32, device=dev)
            self._done = torch.zeros(N, dtype=torch.uint8, device=dev)
            self._removed = torch.zeros(N, dtype=torch.int32, device=dev)
            # the flag plane of the step launches (include/madrl_hip.h, madrl_pursuit_flags_offset): uint8 [N, 4] inside the state buffer;
            # step()'s `done` / `truncated` / `count_overflow` are bool VIEWS of its columns -- no torch kernel runs after the launch
            self._flags = self._state[foff.value:foff.value + 4 * N].view(N, 4)
            self._flag_views = tuple(self._flags[:, k].view(torch.bool) for k in range(3))   # done, truncated, count_overflow
            self._shape_key = shape_key
            self._obs_is_fresh = True   # every element +0.0f, like the reference's local_obs at construction (:119-120)
            self._obs_version = None    # (no launch has written this tensor yet: nothing to compare its version counter with)
        self.obs_dim = D
        self._destroy()
        h = C.c_void_p()
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        _lib.check(L.madrl_pursuit_create(C.byref(cfg), self.map_pool.ctypes.data_as(C.c_void_p), N,
                                          dev_index, _lib.ptr(self._state), C.byref(h)))
        self._handle = h
        self._handle_key = self._create_key()
        if getattr(self, "_obs_is_fresh", False):   # nothing has written the buffer yet: tell the fast path (no cell "unknown" at the start)
            _lib.check(L.madrl_pursuit_declare_obs_zero(h, _lib.ptr(self._obs), _lib.current_stream(self.device)))
        self.handle_generation = getattr(self, "handle_generation", 0) + 1   # how many times the native handle was (re-)created
        if getattr(self, "_cw_env", None) is not None and self._cw_env.shape[0] == N:   # per-env curriculum follows the new handle
            _lib.check(L.madrl_pursuit_set_curriculum(h, _lib.ptr(self._cw_env), _lib.ptr(self._catchr_env)))
        else:
            self._cw_env = self._catchr_env = None
        if self._threads or self._max_blocks:
            _lib.check(L.madrl_pursuit_set_launch(h, self._threads, self._max_blocks))
        if getattr(self, "_kernel", "auto") != "auto":
            self.set_kernel(self._kernel)
        elif N >= 4096:
            self._hint_fast_path()
        if getattr(self, "_walk", "auto") != "auto":
            self.set_walk(self._walk)
        obs_shape = (D,) if self.flatten else (self.obs_range, self.obs_range, 4)
        self.pursuers = [PursuitAgent(obs_shape) for _ in range(P)]
        self.act_dims = [5] * P

    _hinted = set()

    def _hint_fast_path(self):
        """A large batch of a shape that COULD have a compile-time specialised kernel but runs on the generic one (about half the speed): say
        once per shape how to get it (python -m madrl_amd.build --pursuit-shape ...; csrc/pursuit_specializations.def)."""
        from . import build as _build
        shape = (self.xs, self.ys, int(self.n_pursuers), int(self.n_evaders), int(self.obs_range), int(bool(self.flatten)))
        if shape in BatchedPursuitEvade._hinted or self.kernel_kind != "generic" or not self.train_pursuit and shape[2] + shape[3] > 64:
            return
        kind, _ = _build.pursuit_fast_path(*shape, include_id=bool(self.include_id))
        if kind is not None:
            import warnings
            BatchedPursuitEvade._hinted.add(shape)
            warnings.warn("PursuitEvade %dx%d, %d v %d, obs_range %d runs on the generic kernel; `python -m madrl_amd.build --pursuit-shape %s` "
                          "compiles the specialised kernel for this shape (results are identical, a step takes about half the time)"
                          % (shape[0], shape[1], shape[2], shape[3], shape[4], " ".join(str(v) for v in shape)), stacklevel=3)

    def set_kernel(self, kind):
        """'auto' | 'generic' | 'wave' (one wavefront per env, compile-time specialised shapes only)"""
        k = {"auto": _lib.KERNEL_AUTO, "generic": _lib.KERNEL_GENERIC, "wave": _lib.KERNEL_WAVE}[kind]
        _lib.check(_lib.lib().madrl_pursuit_set_kernel(self._handle, k))
        self._kernel = kind

    @property
    def kernel_kind(self):
        out = C.c_int32()
        _lib.check(_lib.lib().madrl_pursuit_kernel_kind(self._handle, C.byref(out)))
        return {_lib.KERNEL_GENERIC: "generic", _lib.KERNEL_WAVE: "wave"}[out.value]

    def set_walk(self, mode):
        """'auto' | 'alternate' | 'forward': the order in which successive launches of the fast path walk the env range (a
        memory-side cache matter for large batches; results do not depend on it)"""
        _lib.check(_lib.lib().madrl_pursuit_set_walk(self._handle, {"auto": 0, "alternate": 1, "forward": 2}[mode]))
        self._walk = mode

    def set_launch(self, threads=0, max_blocks=0):
        self._threads, self._max_blocks = int(threads), int(max_blocks)
        _lib.check(_lib.lib().madrl_pursuit_set_launch(self._handle, self._threads, self._max_blocks))

    def _destroy(self):
        if getattr(self, "_handle", None):
            _lib.lib().madrl_pursuit_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    def _stream(self):
        return _lib.current_stream(self.device)

    def _obs_view(self):
        if self.flatten:
            return self._obs
        R = self.obs_range
        return self._obs.view(self.n_envs, int(self.n_pursuers), R, R, 4)

    def _i32(self, t, shape, name):
        if t is None:
            return None
        t = torch.as_tensor(t, device=self.device)
        if tuple(t.shape) != tuple(shape):
            if t.numel() != int(np.prod(shape)):
                raise ValueError("%s has shape %s, expected %s" % (name, tuple(t.shape), tuple(shape)))
            t = t.reshape(shape)
        return t.to(torch.int32).contiguous()

    # ------------------------------------------------------------------ reference API
    @property
    def agents(self):
        return self.pursuers

    @property
    def reward_mech(self):
        return self._reward_mech

    def seed(self, seed=None):
        """pursuit_evade.py:166-168.  Here the seed keys every in-kernel Philox draw."""
        if seed is None:
            seed = int(np.random.randint(2**31 - 1))
        self._seed_value = int(seed)
        self.setup()
        return [self._seed_value]

    def get_param_values(self):
        return self.__dict__

    def n_agents(self):
        return int(self.n_pursuers)

    def reset(self, mask=None, positions=None, map_ids=None):
        """pursuit_evade.py:173-207 for every env (or those with mask != 0).
        positions int [N, P+E, 2] / map_ids int [N] replace the random draws (parity hook)."""
        N, A = self.n_envs, int(self.n_pursuers) + int(self.n_evaders)
        if mask is not None:
            mask = torch.as_tensor(mask, device=self.device).reshape(N).to(torch.uint8).contiguous()
        pos = self._i32(positions, (N, A, 2), "positions")
        mid = self._i32(map_ids, (N,), "map_ids")
        if mask is not None and getattr(self, "_needs_reset", False):
            raise RuntimeError("the agent counts changed (update_curriculum / set_param_values): the whole batch must be reset() once")
        self._check_obs_untouched()
        _lib.check(_lib.lib().madrl_pursuit_reset(self._handle, _lib.ptr(mask), _lib.ptr(pos), _lib.ptr(mid),
                                                  _lib.ptr(self._obs), self._stream()))
        self._was_reset, self._needs_reset, self._obs_is_fresh = True, False, False
        self._obs_version = self._version_of_obs()
        return self._obs_view()

    def _check_obs_untouched(self):
        """The returned observation tensor IS the persistent IN / OUT buffer (the reference's local_obs, pursuit_evade.py:119-120, whose
        (R, R, 4) views it hands out the same way, :441-449).  The fast path remembers which of its never-stored cells hold 0.0 (stale-zero
        masks); a caller that edits the tensor in place -- `obs.sub_(mean)` -- would make that memory wrong without anybody noticing.
        PyTorch counts in-place operations per tensor (`_version`, shared by every view): if the count moved since this env last wrote the
        buffer, the masks are reset to "nothing known" before the next launch.  The edit itself stays in the stale cells, exactly as an
        edit of local_obs would in the reference; writes PyTorch cannot see (another library writing through the raw pointer) still need
        invalidate_obs()."""
        v = self._version_of_obs()
        if v is None or (getattr(self, "_obs_version", None) is not None and v != self._obs_version):
            _lib.check(_lib.lib().madrl_pursuit_invalidate_obs(self._handle))
            self._obs_is_fresh = False
        self._obs_version = v

    def _version_of_obs(self):
        """the observation tensor's in-place operation count, or None where PyTorch keeps none: a tensor allocated under
        torch.inference_mode() ("Inference tensors do not track version counter") -- edits of such a buffer cannot be noticed, so the
        stale-zero masks are reset before EVERY launch then (correct, the slower "nothing known" state of the fast path).  Under hipGraph
        capture this host-side check runs once, at capture: a replayed graph neither re-checks nor re-invalidates -- do not edit the
        observation buffer in place between replays (or call invalidate_obs() and re-capture)."""
        try:
            return self._obs._version
        except RuntimeError:
            return None

    def step(self, actions, evader_actions=None, rew_out=None, done_out=None):
        """pursuit_evade.py:209-262.  actions: int [N, P] (0..4).  evader_actions: optional int
        [N, E], entry k drives the k-th remaining evader (scripted evader_controller).
        rew_out float32 [N, P] / done_out uint8 [N]: optional contiguous destinations (e.g. a slot of a trajectory tensor) the
        kernel writes instead of the env's own buffers -- the C ABI takes any device pointer, no copy afterwards."""
        N, P, E = self.n_envs, int(self.n_pursuers), int(self.n_evaders)
        if getattr(self, "_needs_reset", False):
            raise RuntimeError("update_curriculum / set_param_values changed the agent counts: the running episodes cannot continue "
                               "(the reference applies new counts at the next reset(), pursuit_evade.py:173-199) -- call reset() first")
        if type(actions) is torch.Tensor and actions.dtype is torch.int32 and actions.shape == (N, P) and actions.device == self.device and actions.is_contiguous():
            act = actions   # the sampler's own tensor, already what the kernel reads
        else:
            act = self._i32(actions, (N, P), "actions")
        # evader control (train_pursuit=False): the opponents are the pursuers, one injected action per pursuer
        eact = self._i32(evader_actions, (N, E if self.train_pursuit else P), "evader_actions")
        self._check_obs_untouched()
        self._obs_is_fresh = False
        rew = self._rew if rew_out is None else rew_out
        dn = self._done if done_out is None else done_out
        assert rew.dtype == torch.float32 and rew.numel() == N * P and dn.dtype == torch.uint8 and dn.numel() == N
        _lib.check(_lib.lib().madrl_pursuit_step(self._handle, _lib.ptr(act), _lib.ptr(eact), _lib.ptr(self._obs),
                                                 _lib.ptr(rew), _lib.ptr(dn), _lib.ptr(self._removed),
                                                 self._stream()))
        return self._step_result(rew, dn)

    def step_into(self, actions, rew_out, done_out):
        """The launch of step() alone, for a sampler loop that keeps everything on the device: rewards and done bits go to the given
        trajectory slots, the observation view is returned, and none of the small torch kernels that build step()'s `done` / `info`
        tensors run (six launches of ~5 us each: a third of a 65 536-env rollout step, profiles/r05_rollout).  actions: int32 [N, P]
        contiguous on the env's device."""
        N, P = self.n_envs, int(self.n_pursuers)
        if getattr(self, "_needs_reset", False):
            raise RuntimeError("update_curriculum / set_param_values changed the agent counts -- call reset() first")
        assert actions.dtype == torch.int32 and actions.is_contiguous() and actions.numel() == N * P
        assert rew_out.dtype == torch.float32 and rew_out.numel() == N * P and done_out.dtype == torch.uint8 and done_out.numel() == N
        self._check_obs_untouched()
        self._obs_is_fresh = False
        _lib.check(_lib.lib().madrl_pursuit_step(self._handle, _lib.ptr(actions), None, _lib.ptr(self._obs), _lib.ptr(rew_out), _lib.ptr(done_out),
                                                 _lib.ptr(self._removed), self._stream()))
        return self._obs_view()

    def _step_result(self, rew, dn):
        """Every returned tensor is a view of a buffer the step launch itself wrote (no torch kernel runs here: the drop-in API costs what
        the C ABI call costs, bench.py `python_api_ms_per_step`) and, like `obs` and `rew`, holds this step's values until the next step().
        bit 7 / count_overflow: more than 253 agents of one kind stood on ONE cell of this env (byte count grids of the generic kernel;
        possible only with more than 253 pursuers or evaders) -- its results are void until its next reset"""
        done, trunc, ovf = self._flag_views
        return self._obs_view(), rew, done, {"removed": self._removed, "truncated": trunc, "done_bits": dn, "count_overflow": ovf}

    def obs_rows_valid(self):
        """bool [N, P]: which observation rows the last reset / step wrote.  All of them with train_pursuit; in evader control
        row k is the k-th remaining evader among slots 0..P-1 (pursuit_evade.py:418-428) and the rows past the last one are
        stale."""
        N, P = self.n_envs, int(self.n_pursuers)
        if self.train_pursuit:
            return torch.ones((N, P), dtype=torch.bool, device=self.device)
        left = (self.get_state()["gone"][:, :P] == 0).sum(dim=1, keepdim=True)
        return torch.arange(P, device=self.device)[None, :] < left

    @property
    def is_terminal(self):
        """pursuit_evade.py:383-389 per env: no evaders left."""
        return self.get_state()["gone"].bool().all(dim=1)

    # ------------------------------------------------------------------ curriculum (pursuit_evade.py:264-272)
    def _create_key(self):
        """everything madrl_pursuit_create bakes into the handle except catchr / constraint_window (those two can be
        changed in place: madrl_pursuit_set_params)"""
        c = self._config()
        return tuple(getattr(c, n) for n, _ in c._fields_ if n not in ("catchr", "constraint_window")) + (self.map_pool.tobytes(),)

    def set_param_values(self, lut):
        """madrl_environments/__init__.py:64-67: setattr + setup().  The native handle is only re-created when something other
        than catchr / constraint_window changed."""
        for k, v in lut.items():
            if k == "reward_mech":
                self._reward_mech = v
            else:
                setattr(self, k, v)
        self._apply_params()

    def _apply_params(self):
        if self._handle is not None and self._create_key() == getattr(self, "_handle_key", None):
            _lib.check(_lib.lib().madrl_pursuit_set_params(self._handle, float(self.catchr), float(self.constraint_window)))
        else:
            # Something the handle bakes in changed (agent counts, map, observation shape): the handle is re-created, and when the
            # state layout changed with it the state starts from the constructor's (every agent at (0, 0)).  In the reference the
            # running episode goes on with the old agents and the new counts take effect at the next reset() (:173-199); here the
            # running episodes cannot continue, so step() refuses until reset() has been called.
            had_episode, key = getattr(self, "_was_reset", False), getattr(self, "_shape_key", None)
            self.setup()
            if had_episode and key != self._shape_key:
                self._needs_reset = True
            if self._cw_env is not None:   # the per-env curriculum arrays belong to the env object, not to the handle
                _lib.check(_lib.lib().madrl_pursuit_set_curriculum(self._handle, _lib.ptr(self._cw_env), _lib.ptr(self._catchr_env)))

    @staticmethod
    def curriculum_next(itr, constraint_window, n_evaders, n_pursuers, catchr, constrain_rate, remove_every, turn_off_shaping):
        """One PursuitEvade.update_curriculum(itr) on plain values (pursuit_evade.py:264-272): returns the new
        (constraint_window, n_evaders, n_pursuers, catchr)."""
        constraint_window = constraint_window + constrain_rate      # :265
        constraint_window = np.clip(constraint_window, 0.0, 1.0)    # :266
        if itr != 0 and itr % remove_every == 0 and n_pursuers > 4:  # :268-270
            n_evaders -= 1
            n_pursuers -= 1
        if itr > turn_off_shaping:                                   # :271-272
            catchr = 0.0
        return constraint_window, n_evaders, n_pursuers, catchr

    def update_curriculum(self, itr, mask=None):
        """pursuit_evade.py:264-272.  Without `mask`: the whole batch moves one curriculum iteration, like the reference object
        (the handle is re-created only when the agent counts change).  With `mask` (bool / uint8 [N]): only those env
        instances advance -- constraint_window and catchr become PER-ENV device arrays (`curriculum_state()`), read by the
        kernels in place; the agent counts of a batch cannot differ per env, so the remove-agents rule is not applied then."""
        if mask is None and self._cw_env is None:
            cw, ne, np_, cr = self.curriculum_next(itr, self.constraint_window, self.n_evaders, self.n_pursuers, self.catchr,
                                                   self.curriculum_constrain_rate, self.curriculum_remove_every,
                                                   self.curriculum_turn_off_shaping)
            self.constraint_window, self.n_evaders, self.n_pursuers, self.catchr = cw, ne, np_, cr
            self._apply_params()
            return
        self._bind_curriculum()
        m = torch.ones(self.n_envs, dtype=torch.bool, device=self.device) if mask is None else \
            torch.as_tensor(mask, device=self.device).reshape(self.n_envs).bool()
        # the same float64 operations as :265-266 / :271-272, on the masked elements
        cw = torch.clamp(self._cw_env + float(self.curriculum_constrain_rate), 0.0, 1.0)
        self._cw_env.copy_(torch.where(m, cw, self._cw_env))
        if itr > self.curriculum_turn_off_shaping:
            self._catchr_env.copy_(torch.where(m, torch.zeros_like(self._catchr_env), self._catchr_env))

    def _bind_curriculum(self):
        if self._cw_env is None:
            f64 = dict(dtype=torch.float64, device=self.device)
            self._cw_env = torch.full((self.n_envs,), float(self.constraint_window), **f64)
            self._catchr_env = torch.full((self.n_envs,), float(self.catchr), **f64)
            _lib.check(_lib.lib().madrl_pursuit_set_curriculum(self._handle, _lib.ptr(self._cw_env), _lib.ptr(self._catchr_env)))

    def set_curriculum(self, constraint_window=None, catchr=None):
        """Per-env curriculum values: float64 [N] (anything that converts).  They stay bound until clear_curriculum()."""
        self._bind_curriculum()
        if constraint_window is not None:
            cw = torch.as_tensor(constraint_window, dtype=torch.float64, device=self.device).reshape(self.n_envs)
            if not bool(((cw > 0) & (cw <= 1)).all()):
                raise ValueError("constraint_window must be in (0, 1]")
            self._cw_env.copy_(cw)
        if catchr is not None:
            self._catchr_env.copy_(torch.as_tensor(catchr, dtype=torch.float64, device=self.device).reshape(self.n_envs))

    def clear_curriculum(self):
        self._cw_env = self._catchr_env = None
        _lib.check(_lib.lib().madrl_pursuit_set_curriculum(self._handle, None, None))

    def curriculum_state(self):
        """(constraint_window, catchr) as float64 [N] tensors (per-env arrays when bound, else the batch-wide scalars)"""
        if self._cw_env is not None:
            return self._cw_env, self._catchr_env
        f64 = dict(dtype=torch.float64, device=self.device)
        return torch.full((self.n_envs,), float(self.constraint_window), **f64), torch.full((self.n_envs,), float(self.catchr), **f64)

    def invalidate_obs(self):
        """Call after writing into `obs_buffer` through something PyTorch does not see (include/madrl_hip.h, obs_dev contract); in-place
        tensor operations are noticed without it (_check_obs_untouched)."""
        _lib.check(_lib.lib().madrl_pursuit_invalidate_obs(self._handle))
        self._obs_is_fresh = False   # a re-created handle must not declare this buffer all-zero

    # ------------------------------------------------------------------ state exchange
    def get_state(self):
        N, P, E, dev = self.n_envs, int(self.n_pursuers), int(self.n_evaders), self.device
        st = dict(pos_p=torch.zeros((N, P, 2), dtype=torch.int32, device=dev),
                  pos_e=torch.zeros((N, E, 2), dtype=torch.int32, device=dev),
                  gone=torch.zeros((N, E), dtype=torch.uint8, device=dev),
                  term_p=torch.zeros((N, P), dtype=torch.uint8, device=dev),
                  term_e=torch.zeros((N, E), dtype=torch.uint8, device=dev),
                  map_id=torch.zeros(N, dtype=torch.int32, device=dev),
                  tick=torch.zeros(N, dtype=torch.int32, device=dev),
                  t=torch.zeros(N, dtype=torch.int32, device=dev))
        _lib.check(_lib.lib().madrl_pursuit_get_state(
            self._handle, *[_lib.ptr(st[k]) for k in ("pos_p", "pos_e", "gone", "term_p", "term_e", "map_id", "tick", "t")],
            self._stream()))
        return st

    def set_state(self, st):
        """Any subset of the get_state() keys.  (The reference's equivalent is poking
        AgentLayer.set_position, pursuit/test_pursuit.py:22-51.)"""
        N, P, E = self.n_envs, int(self.n_pursuers), int(self.n_evaders)
        spec = (("pos_p", (N, P, 2), torch.int32), ("pos_e", (N, E, 2), torch.int32),
                ("gone", (N, E), torch.uint8), ("term_p", (N, P), torch.uint8),
                ("term_e", (N, E), torch.uint8), ("map_id", (N,), torch.int32),
                ("tick", (N,), torch.int32), ("t", (N,), torch.int32))
        args = []
        for k, shape, dt in spec:
            v = st.get(k)
            if v is not None:
                v = torch.as_tensor(np.asarray(v) if not torch.is_tensor(v) else v, device=self.device)
                v = v.reshape(shape).to(dt).contiguous()
            args.append(v)
        self._keepalive = args
        _lib.check(_lib.lib().madrl_pursuit_set_state(self._handle, *[_lib.ptr(a) for a in args], self._stream()))
        self._obs_is_fresh = False

    @property
    def obs_buffer(self):
        """the persistent IN/OUT observation tensor (reference: self.local_obs).  Handing it out ends the "freshly zeroed" promise a
        re-created handle would otherwise make for it (the caller may write through it)."""
        self._obs_is_fresh = False
        return self._obs

    # ------------------------------------------------------------------ pickling (EzPickle-style)
    def __getstate__(self):
        d = dict(self._ctor)
        # curriculum attributes travel with the pickle (pursuit_evade.py:397-411)
        d["curriculum"] = dict(constraint_window=self.constraint_window, n_evaders=self.n_evaders,
                               n_pursuers=self.n_pursuers, catchr=self.catchr)
        if self._cw_env is not None:   # the per-env curriculum (update_curriculum(itr, mask=) / set_curriculum) travels too
            d["curriculum_env"] = dict(constraint_window=self._cw_env.cpu().numpy(), catchr=self._catchr_env.cpu().numpy())
        return d

    def __setstate__(self, d):
        cur = d.pop("curriculum", {})
        cur_env = d.pop("curriculum_env", None)
        kwargs = d.pop("kwargs")
        kwargs.update(cur)
        self.__init__(d.pop("map_pool"), **d, **kwargs)
        if cur_env is not None:
            self.set_curriculum(**cur_env)


class PursuitEvade(SingleEnvDelegate, AbstractMAEnv):
    """N == 1 drop-in with the reference's return types (pursuit_evade.py:26).

    `evader_controller=` (pursuit_evade.py:87): an object with `.act(model_state)`; as in the reference (:238-241) it is
    asked once per REMAINING evader, in layer order, every step, and sees the model_state of the previous step
    (float32 [4, xs, ys]: map, pursuer counts, evader counts incl. the evaders caught in that step, zeros).  Its
    answers travel to the kernel as injected evader actions.  Without it the evaders move by in-kernel Philox draws
    (the reference's default is an unseeded RandomPolicy, Controllers.py:11)."""

    def __init__(self, map_pool, device="cuda:0", **kwargs):
        self._evader_controller = kwargs.pop("evader_controller", None)
        if not kwargs.get("train_pursuit", True):  # the opponents are the pursuers then (:220-224)
            self._evader_controller = kwargs.pop("pursuer_controller", None)
        else:
            kwargs.pop("pursuer_controller", None)  # never asked with train_pursuit (:215-219)
        self._reset_positions = []   # parity hook: see script_reset_positions()
        self._alive_at_step_start = None
        self._env = BatchedPursuitEvade(map_pool, n_envs=1, device=device, **kwargs)

    def script_reset_positions(self, positions):
        """Parity hook (the golden generators replace agent_utils.feasible_position the same way): each following
        reset() takes its initial positions, int [P+E, 2] pursuers first, from this list instead of sampling."""
        self._reset_positions = [np.asarray(p, dtype=np.int32) for p in positions]

    @property
    def model_state(self):
        """pursuit_evade.py:152, :201-203, :244-246 rebuilt from the packed state (host side, N == 1)."""
        e = self._env
        st = {k: v[0].cpu().numpy() for k, v in e.get_state().items()}
        ms = np.zeros((4, e.xs, e.ys), dtype=np.float32)
        ms[0] = e.map_pool[int(st["map_id"])]
        np.add.at(ms[1], (st["pos_p"][:, 0], st["pos_p"][:, 1]), 1)
        drawn = (st["gone"] == 0) if self._alive_at_step_start is None else self._alive_at_step_start  # Q6
        pe = st["pos_e"][drawn]
        np.add.at(ms[2], (pe[:, 0], pe[:, 1]), 1)
        return ms

    @property
    def agents(self):
        return self._env.agents

    @property
    def reward_mech(self):
        return self._env.reward_mech

    def seed(self, seed=None):
        return self._env.seed(seed)

    def set_param_values(self, lut):
        self._env.set_param_values(lut)

    def update_curriculum(self, itr):
        self._env.update_curriculum(itr)

    def _obslist(self, obs):
        o = obs[0].detach().cpu().numpy().astype(np.float64)
        if self._env.train_pursuit:
            return [o[i] for i in range(o.shape[0])]
        # evader control: collect_obs (:418-428) returns None for a gone evader slot, else the next row
        gone = self._env.get_state()["gone"][0, :o.shape[0]].cpu().numpy()
        out, k = [], 0
        for i in range(o.shape[0]):
            if gone[i]:
                out.append(None)
            else:
                out.append(o[k]); k += 1
        return out

    def reset(self):
        pos = self._reset_positions.pop(0)[None] if self._reset_positions else None
        self._alive_at_step_start = None
        return self._obslist(self._env.reset(positions=pos))

    def step(self, actions):
        P, E = int(self._env.n_pursuers), int(self._env.n_evaders)
        if isinstance(actions, (list, np.ndarray)):  # pursuit_evade.py:227-230
            act = np.asarray(actions).reshape(-1)
            if act.shape[0] != P:
                raise ValueError("expected %d actions, got %d" % (P, act.shape[0]))
        else:  # joint scalar action, :231-235
            act = np.asarray(np.unravel_index(int(actions), self._env.act_dims))
        if ((act < -5) | (act > 4)).any():
            raise IndexError("list index out of range")  # motion_range[a], DiscreteAgent.py:83
        act = np.where(act < 0, act + 5, act)  # python list wrap-around
        eact = None
        if self._evader_controller is not None:  # :238-241
            ms = self.model_state
            alive = self._env.get_state()["gone"][0].cpu().numpy() == 0
            if self._env.train_pursuit:
                eact = np.full((1, E), 4, dtype=np.int32)
                for k in range(int(alive.sum())):
                    eact[0, k] = int(self._evader_controller.act(ms))
            else:  # one pursuer_controller.act per pursuer
                eact = np.array([[int(self._evader_controller.act(ms)) for _ in range(P)]], dtype=np.int32)
            self._alive_at_step_start = alive
        obs, rew, done, info = self._env.step(torch.as_tensor(act.reshape(1, P)), evader_actions=eact)
        if bool(info["count_overflow"][0].item()):
            raise OverflowError("more than 253 agents of one kind on one cell: outside the byte count grids of the kernel")
        r = rew[0].detach().cpu().numpy().astype(np.float64)
        rewards = [float(r[0])] * P if self._env.reward_mech == "global" else r
        return self._obslist(obs), rewards, bool(done[0].item()), {"removed": int(info["removed"][0].item())}

    @property
    def is_terminal(self):
        return bool(self._env.is_terminal[0].item())
