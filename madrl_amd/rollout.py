"""GPU rollout collector: the loop the reference leaves to its external samplers
(`parallel_sampler` runners/rurllab.py:259, `ParallelSampler`/`SimpleSampler` runners/rurltools.py:170-194;
in-tree instance of the same loop: heuristics/pursuit.py:71-85) run over a batched env with the policy in
the loop, producing time-major trajectory tensors [T, N, A, ...] that stay in HBM, plus the
discounted-return / GAE post-processing those samplers apply (runners/rurllab.py:298-305 `discount`,
`gae_lambda`; runners/rurltools.py:196-209) as one reverse-scan kernel (madrl_rollout_gae).

The env must be an `auto_reset` batched env (episode boundaries are the done flags; the observation
returned by a done step already belongs to the next episode) or be shorter-lived than the horizon.
"""
import numpy as np
import torch

from . import _lib


class Trajectory(object):
    """time-major tensors of one collect() call: actions [T,N,A(,adim)], rewards float32 [T,N,A],
    dones uint8 [T,N] (bit0 terminal, bit1 time limit), values float32 [T+1,N,A] or None,
    observations float32 [T,N,A,D] or None, returns / advantages float32 [T,N,A]."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def paths(self, env_ids=None):
        """The list-of-path-dicts view of the reference's samplers (one dict per finished or cut episode segment
        and agent: "observations", "actions", "rewards", "returns", "advantages"), built on the host for the
        chosen env instances (default: all; meant for small N / debugging, the tensors are the product)."""
        T, N = self.dones.shape
        ids = range(N) if env_ids is None else env_ids
        dn = self.dones.cpu().numpy() != 0
        host = {k: getattr(self, k).cpu().numpy() for k in ("actions", "rewards", "returns") }
        for k in ("advantages", "observations"):
            if getattr(self, k) is not None:
                host[k] = getattr(self, k).cpu().numpy()
        out = []
        for n in ids:
            cuts = [t + 1 for t in range(T) if dn[t, n]]
            if not cuts or cuts[-1] != T:
                cuts.append(T)
            s = 0
            for e in cuts:
                for a in range(self.rewards.shape[2]):
                    d = {k: v[s:e, n, a] for k, v in host.items()}
                    d["env_id"], d["agent_id"], d["terminated"] = n, a, bool(dn[e - 1, n])
                    out.append(d)
                s = e
        return out


class RolloutCollector(object):
    """collector = RolloutCollector(env, policy, horizon); traj = collector.collect()

    policy(obs [N,A,D]) -> actions [N,A(,adim)] or (actions, values [N,A]); it runs on the env's device.
    State (the current observation) carries over between collect() calls, like a sampler that keeps its
    env copies alive between iterations."""

    def __init__(self, env, policy, horizon, discount=0.99, gae_lambda=1.0, store_observations=False, graph=False):
        """graph=True: from the second collect() on, the whole horizon (policy launches, step kernels, buffer copies, the
        return scan) is one captured hipGraph that is replayed -- for small batches the per-launch overhead of ~4 launches
        per step otherwise dominates.  The policy must be capturable (no host-side state that changes per call, no syncs);
        the device policies of madrl_amd.heuristics are."""
        self.env, self.policy, self.T = env, policy, int(horizon)
        self.discount, self.gae_lambda, self.store_observations = float(discount), float(gae_lambda), store_observations
        self._obs = None
        self._buf = None
        self._use_graph, self._graph, self._calls = bool(graph), None, 0
        import inspect
        self._direct = "rew_out" in inspect.signature(env.step).parameters   # envs whose step() can write into caller buffers
        self._into = hasattr(env, "step_into")                               # ... and that offer the launch alone (no done / info tensors built)
        try:
            self._policy_out = "out" in inspect.signature(policy).parameters   # policies that write into a trajectory slot
        except (TypeError, ValueError):
            self._policy_out = False

    def _alloc(self, obs, act, val):
        T, dev = self.T, obs.device
        b = dict(actions=torch.empty((T,) + tuple(act.shape), dtype=act.dtype, device=dev),
                 rewards=torch.empty((T,) + tuple(obs.shape[:2]), dtype=torch.float32, device=dev),
                 dones=torch.empty((T, obs.shape[0]), dtype=torch.uint8, device=dev),
                 returns=torch.empty((T,) + tuple(obs.shape[:2]), dtype=torch.float32, device=dev))
        b["values"] = torch.empty((T + 1,) + tuple(obs.shape[:2]), dtype=torch.float32, device=dev) if val is not None else None
        b["advantages"] = torch.empty_like(b["returns"]) if val is not None else None
        b["observations"] = torch.empty((T,) + tuple(obs.shape), dtype=torch.float32, device=dev) if self.store_observations else None
        return b

    def _act(self, obs):
        out = self.policy(obs)
        return out if isinstance(out, tuple) else (out, None)

    def collect(self):
        self._calls += 1
        if not self._use_graph or self._calls == 1:  # the first call also warms up every allocation
            return self._collect_eager()
        if self._graph is None:
            torch.cuda.synchronize(self._obs.device)
            self._graph = torch.cuda.CUDAGraph()
            self._obs_in = self._obs
            with torch.cuda.graph(self._graph):  # records, does not execute
                self._collect_eager()
                self._obs_in.copy_(self._obs)    # the env returns the same persistent tensor; keep it explicit
        self._graph.replay()
        return Trajectory(**self._buf)

    def _collect_eager(self):
        self._begin()
        for t in range(self.T):
            self._step(t)
        return self._finish()

    # the three parts of one horizon; ShardedRolloutCollector interleaves the _step(t) of several collectors on their own streams
    def _begin(self):
        if self._obs is None:
            self._obs = self.env.reset()

    def _step(self, t):
        env, obs = self.env, self._obs
        if self._policy_out and self._buf is not None:   # the policy writes its actions straight into their trajectory slot
            out = self.policy(obs, out=self._buf["actions"][t])
            act, val = out if isinstance(out, tuple) else (out, None)   # (a policy with `out=` may return values too: they are stored below)
            if act.data_ptr() != self._buf["actions"][t].data_ptr():    # ... or ignore `out` and return its own tensor
                self._buf["actions"][t].copy_(act)
        else:
            act, val = self._act(obs)
            if self._buf is None:
                self._buf = self._alloc(obs, act, val)
            self._buf["actions"][t].copy_(act)
        b = self._buf
        if b["observations"] is not None:
            b["observations"][t].copy_(obs)
        if val is not None:
            b["values"][t].copy_(val)
        if self._into and act.dtype == torch.int32 and act.is_contiguous():
            obs = env.step_into(act, b["rewards"][t], b["dones"][t])
        elif self._direct:   # the step kernel writes rewards / done bits straight into their trajectory slot
            obs, rew, done, info = env.step(act, rew_out=b["rewards"][t], done_out=b["dones"][t])
        else:
            obs, rew, done, info = env.step(act)
            b["rewards"][t].copy_(rew)
            b["dones"][t].copy_(info["done_bits"] if isinstance(info, dict) and "done_bits" in info else done.to(torch.uint8))
        self._obs = obs

    def _finish(self):
        obs, b = self._obs, self._buf
        if b["values"] is not None:
            b["values"][self.T].copy_(self._act(obs)[1])  # bootstrap of the unfinished tail
        N, A = b["rewards"].shape[1:]
        _lib.check(_lib.lib().madrl_rollout_gae(_lib.ptr(b["rewards"]), _lib.ptr(b["dones"]),
                                                _lib.ptr(b["values"]) if b["values"] is not None else None, self.T, N, A,
                                                self.discount, self.gae_lambda, _lib.ptr(b["returns"]),
                                                _lib.ptr(b["advantages"]) if b["advantages"] is not None else None,
                                                _lib.current_stream(obs.device)))
        return Trajectory(**b)


class ShardedRolloutCollector(object):
    """The same rollout over a `StreamSharded` env (madrl_amd/sharded.py): one RolloutCollector per sub-batch, each driven on its
    sub-batch's HIP stream, their steps interleaved so that every stream always has work queued.  Sub-batch A's policy launch then runs
    under sub-batch B's step kernel -- the double-buffered sampler.  `policies`: one policy object per sub-batch (policies with device
    state, like the heuristic chase policy's draw counter, must not be shared).
    collect() -> list of Trajectory, one per sub-batch (rows [j * per, (j + 1) * per) of the batch), after joining the caller's stream."""

    def __init__(self, sharded_env, policies, horizon, **kw):
        assert len(policies) == sharded_env.n_streams
        self.sharded = sharded_env
        self.collectors = [RolloutCollector(e, p, horizon, **kw) for e, p in zip(sharded_env.envs, policies)]
        self.T = int(horizon)

    def collect(self):
        sh = self.sharded
        sh.fork()
        if self.collectors[0]._use_graph:
            # one captured hipGraph per sub-batch and horizon, replayed on the sub-batch's stream: the host issues S launches per
            # horizon instead of ~10 per step and sub-batch (the eager interleaving below is bound by the host's launch rate at two
            # sub-batches already: 141 us per step against 139 for the single collector at 65 536 envs)
            out = []
            for c, s in zip(self.collectors, sh.streams):
                with torch.cuda.stream(s):
                    out.append(c.collect())
            sh.join()
            return out
        for c, s in zip(self.collectors, sh.streams):
            with torch.cuda.stream(s):
                c._begin()
        for t in range(self.T):
            for c, s in zip(self.collectors, sh.streams):
                with torch.cuda.stream(s):
                    c._step(t)
        out = []
        for c, s in zip(self.collectors, sh.streams):
            with torch.cuda.stream(s):
                out.append(c._finish())
        sh.join()
        return out
