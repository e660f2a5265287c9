"""One env batch stepped as S independent sub-batches, each on its own HIP stream.

Env instances never interact (the reference's own parallelism is N pickled env copies in sampler workers, runners/rurllab.py:259,
runners/rurltools.py:184-191), so nothing orders sub-batch A's step t + 1 against sub-batch B's step t.  A kernel launch begins with
every wavefront in lockstep and ends with a drain in which the last wavefronts finish alone; on ONE stream the next launch starts only
after that drain.  With the batch cut into sub-batches on their own streams, one sub-batch's drain overlaps another's ramp-up, and a
VALU-bound simulation launch of one overlaps a bandwidth-bound wrapper launch of another -- the same trick a double-buffered sampler
plays with policy inference.  Measured on MI355X: PursuitEvade 65 536 envs 71-74 us per step as one launch, 60 us as
two sub-batches.

Results do not depend on the sharding: sub-batch j is created with env_id_base advanced by its offset, the same mechanism that makes
multi-GPU sharding invisible (tests/test_sharded_gpu.py)."""
import torch

_POOL = {}


def shared_streams(device, n):
    """The process-wide sub-batch streams of `device`: the first `n` of a list that only ever grows.  HIP multiplexes its streams onto
    a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default) in creation order, and two sub-batch streams that end up on ONE queue run
    their launches one after the other -- measured: Waterworld's two sub-batches 61 instead of 42 us per step when they were the third
    and fourth stream a process had created; MultiWalker's four sub-batches 6.7 instead of 3.4 ms.  So every StreamSharded (and bench.py)
    uses the same few streams, created and bound together; with more than two sub-batches next to other stream users, start the process
    with GPU_MAX_HW_QUEUES=8 (read by the HIP runtime at start-up; bench.py sets it before importing torch)."""
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    pool = _POOL.setdefault(key, [])
    while len(pool) < max(n, 4):
        s = torch.cuda.Stream(device=device)
        with torch.cuda.stream(s):     # first use binds the stream to its hardware queue: all of them now, in this order, whatever the
            torch.zeros(1, device=device)   # process does later (a pair created on demand, after other streams had run, measured 61 / 42 us)
        pool.append(s)
    return pool[:n]


class StreamSharded(object):
    def __init__(self, make_env, n_envs, n_streams=2, env_id_base=0, device="cuda:0"):
        """make_env(n_envs=, env_id_base=, device=) -> a Batched* env (or a wrapper around one).  n_envs must divide by n_streams."""
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("StreamSharded needs a ROCm device (HIP streams)")
        if n_streams < 1 or n_envs % n_streams:
            raise ValueError("n_envs=%d does not divide into %d sub-batches" % (n_envs, n_streams))
        self.n_envs, self.n_streams, self.per = int(n_envs), int(n_streams), int(n_envs) // int(n_streams)
        self.envs = [make_env(n_envs=self.per, env_id_base=int(env_id_base) + j * self.per, device=self.device) for j in range(self.n_streams)]
        self.streams = shared_streams(self.device, self.n_streams)
        self._nat = None

    @classmethod
    def from_envs(cls, envs, device=None):
        """the same over env objects that exist already (equal sizes, env_id_base advanced by the caller), e.g. bench.py's sub-batches"""
        self = cls.__new__(cls)
        self.device = torch.device(device if device is not None else envs[0].device)
        self.envs, self.n_streams, self.per = list(envs), len(envs), int(envs[0].n_envs)
        self.n_envs = self.per * self.n_streams
        self.streams = shared_streams(self.device, self.n_streams)
        self._nat = None
        return self

    @property
    def agents(self):
        return self.envs[0].agents

    @property
    def reward_mech(self):
        return self.envs[0].reward_mech

    def _split(self, t):
        if isinstance(t, (list, tuple)):
            assert len(t) == self.n_streams
            return list(t)
        return [t[j * self.per:(j + 1) * self.per] for j in range(self.n_streams)]

    def fork(self):
        """every sub-batch stream waits for what the caller's stream has enqueued so far (e.g. the actions it is about to read)"""
        cur = torch.cuda.current_stream(self.device)
        for s in self.streams:
            s.wait_stream(cur)

    def join(self):
        """the caller's stream waits for every sub-batch (before it reads observations / rewards of all of them)"""
        cur = torch.cuda.current_stream(self.device)
        for s in self.streams:
            cur.wait_stream(s)

    def reset(self, join=True, **kw):
        """-> list of the sub-batches' observation tensors (row j = envs [j * per, (j + 1) * per))"""
        self.fork()
        out = []
        for env, s in zip(self.envs, self.streams):
            with torch.cuda.stream(s):
                out.append(env.reset(**kw))
        if join:
            self.join()
        self._nat = None
        return out

    def step(self, actions, fork=True, join=True, **kw):
        """actions: one tensor [N, ...] (split by rows) or a list with one tensor per sub-batch.  fork / join: order the sub-batch
        streams after / before the caller's stream.  A double-buffered sampler drives each sub-batch from its own stream and needs
        neither; `join=False` returns at once and `join()` is called when all results are wanted (the action tensors are then
        registered with the sub-batch streams, so the caching allocator does not hand their memory out while a launch still reads it).
        -> list of (obs, rew, done, info), one per sub-batch"""
        parts = self._split(actions)
        if not kw and self._native_pursuit():
            return self._step_pursuit_native(parts, fork, join)
        if fork:
            self.fork()
        out = []
        for env, s, a in zip(self.envs, self.streams, parts):
            # envs whose step() is one launch and nothing else take the stream as an argument: entering a stream context costs the host
            # more than the launch (hostage world, two sub-batches: 34 us per step with the contexts against a 26 us kernel)
            on = getattr(env, "step_on_stream", None)
            r = on(a, s) if (on is not None and not kw) else None
            if r is None:
                with torch.cuda.stream(s):
                    r = env.step(a, **kw)
            out.append(r)
            if not join and torch.is_tensor(a):
                a.record_stream(s)
        if join:
            self.join()
        return out

    def _native_pursuit(self):
        from .pursuit import BatchedPursuitEvade
        return all(type(e) is BatchedPursuitEvade for e in self.envs)

    def _step_pursuit_native(self, parts, fork, join):
        """PursuitEvade sub-batches: ONE call of the C ABI forks, launches every sub-batch on its stream and joins
        (madrl_pursuit_step_sharded) -- what fork() / step() x S / join() above do from Python with five event calls per step, at which the
        host, not the GPU, sets the pace (93 us per joined step of 65 536 envs against 60 us for the kernels)."""
        import ctypes as C
        from . import _lib
        S = self.n_streams
        if getattr(self, "_nat", None) is None:
            io = (_lib.PursuitShardIO * S)()
            for j, (e, st) in enumerate(zip(self.envs, self.streams)):
                io[j].inj_evader_actions = None
                io[j].obs, io[j].rew, io[j].done, io[j].removed = (_lib.ptr(t).value for t in (e._obs, e._rew, e._done, e._removed))
                io[j].stream = st.cuda_stream
            self._nat = (io, [e.handle_generation for e in self.envs])
        io, gens = self._nat
        acts = []
        for j, (e, a) in enumerate(zip(self.envs, parts)):
            if getattr(e, "_needs_reset", False) or e.handle_generation != gens[j]:
                self._nat = None   # the handle was re-created (curriculum): take the general path this once, rebuild the table next time
                return self._step_general(parts, fork, join)
            # Nothing of this path may run on the CALLER's stream unordered against the sub-batch launches.  Actions that need a
            # conversion kernel (another dtype / device / layout) go through the general path, which converts on the sub-batch's own
            # stream; step()'s result tensors are views of buffers the launch itself writes (BatchedPursuitEvade._step_result: no torch
            # kernel), so with join=False they hold the step's values exactly when the sub-batch stream has run the launch.
            if not (torch.is_tensor(a) and a.dtype == torch.int32 and a.device == e.device and a.is_contiguous()
                    and a.numel() == e.n_envs * int(e.n_pursuers)):
                return self._step_general(parts, fork, join)
            acts.append(a)
        for j, (e, a) in enumerate(zip(self.envs, acts)):
            e._check_obs_untouched()
            e._obs_is_fresh = False
            io[j].actions = _lib.ptr(a).value
            if not join:
                a.record_stream(self.streams[j])
        hs = (C.c_void_p * S)(*[e._handle.value for e in self.envs])
        _lib.check(_lib.lib().madrl_pursuit_step_sharded(hs, io, S, _lib.current_stream(self.device), int(bool(fork)), int(bool(join))))
        self._keep = acts
        return [e._step_result(e._rew, e._done) for e in self.envs]

    def _step_general(self, parts, fork, join):
        if fork:
            self.fork()
        out = []
        for env, s, a in zip(self.envs, self.streams, parts):
            with torch.cuda.stream(s):
                out.append(env.step(a))
            if not join and torch.is_tensor(a):
                a.record_stream(s)
        if join:
            self.join()
        return out

    def synchronize(self):
        for s in self.streams:
            s.synchronize()
