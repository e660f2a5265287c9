"""Multi-GPU layer: env instances never interact (each reference env object is self-contained;
the reference's own parallelism is N pickled copies handed to sampler workers,
runners/rurllab.py:259, runners/rurltools.py:184-191), so the path shards by env index with
NO data-path collective.  The single exchange is the episode-end gather of the compact
trajectory (actions, rewards, dones, per-episode info) -- the functional analogue of sampler
workers returning their paths -- done with one all_gather per tensor over RCCL/xGMI
(backend "nccl" on ROCm; "gloo" in the CPU tests) -- or, the default of bench.py since round 6, one gather to the rank that
learns (ChunkedTrajectoryGather mode "root"), or nothing but the per-episode statistics (mode "stats").

Observations are deliberately NOT gathered (C5: ~155 GB per GPU per episode); the policy that
consumes them is data-parallel on the same GPU.
"""
import torch
import torch.distributed as dist


def shard_range(n_total, rank=None, world=None):
    """Contiguous env index range [lo, hi) owned by `rank` (remainder spread over low ranks)."""
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    base, rem = divmod(int(n_total), world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_ragged(local, group=None):
    """local: dict name -> tensor whose LEADING dimension may differ between ranks (uneven env shards from
    shard_range() when n_total % world != 0, per-rank episode counts); trailing dimensions and dtypes agree.
    Returns dict name -> list of `world` tensors, entry r = rank r's tensor.  all_gather_into_tensor needs equal
    shapes, so the leading sizes are exchanged first (one tiny collective for all names), every tensor is padded
    to the largest and the result trimmed."""
    names = sorted(local)
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return {k: [local[k]] for k in names}
    world = dist.get_world_size(group)
    dev = local[names[0]].device
    mine = torch.tensor([int(local[k].shape[0]) for k in names], dtype=torch.int64, device=dev)
    sizes = torch.empty((world, len(names)), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes.view(-1), mine, group=group)
    sizes = sizes.cpu()
    out = {}
    for j, k in enumerate(names):
        v = local[k].contiguous()
        m = int(sizes[:, j].max())
        pad = torch.zeros((m,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
        pad[:v.shape[0]] = v
        buf = torch.empty((world,) + tuple(pad.shape), dtype=v.dtype, device=v.device)
        dist.all_gather_into_tensor(buf.view(-1), pad.view(-1), group=group)
        out[k] = [buf[r, :int(sizes[r, j])] for r in range(world)]
    return out


def gather_trajectories(local, group=None):
    """local: dict name -> tensor with IDENTICAL shape/dtype on every rank (equal env shards: n_total % world == 0;
    use gather_ragged() otherwise -- a shape mismatch here would hang or corrupt the collective, RCCL does not check).
    Returns dict name -> tensor [world, *shape] (every rank gets everything: the learner is
    data-parallel too).  One collective per tensor, issued back to back on the current stream."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return {k: v.unsqueeze(0) for k, v in local.items()}
    world = dist.get_world_size(group)
    out = {}
    for k in sorted(local):
        v = local[k].contiguous()
        buf = torch.empty((world,) + tuple(v.shape), dtype=v.dtype, device=v.device)
        dist.all_gather_into_tensor(buf.view(-1), v.view(-1), group=group)
        out[k] = buf
    return out


class ChunkedTrajectoryGather(object):
    """Overlaps the trajectory exchange with stepping.

    A rollout of T steps is cut into chunks; as soon as a chunk's compact trajectory
    (actions / rewards / dones tensors of shape [chunk, ...]) is complete its all-gather is
    issued with async_op=True -- RCCL runs it on its own HIP stream over xGMI while the step
    kernels of the next chunk run on the compute stream -- and `finish()` waits for all of them
    at episode end.  With ~95 us per 65 536-env step a rank produces ~28 GB/s of trajectory, so a
    single blocking gather at the end would cost ~30 % of the rollout; chunking hides it.
    """

    MODES = ("root", "all", "stats")

    def __init__(self, group=None, always_collective=False, mode="all", dst=0):
        """always_collective: issue the collectives even in a one-rank group (they are then copies made by the backend) instead of
        returning views -- how a one-GPU box exercises the RCCL path (bench.py, MADRL_BENCH_FORCE_COLLECTIVE=1).
        mode: who receives the chunks --
          "root"   rank `dst` only (dist.gather: every other rank just sends; north_star's "gather of trajectory buffers ... at episode
                   end" read as the sampler workers returning their paths to ONE learner process, runners/rurllab.py:259);
          "all"    every rank (all_gather: a learner that is data-parallel over the same ranks; each rank then also RECEIVES world - 1
                   shards -- at configs[1]'s rate about 7 x 42 GB/s of extra HBM writes per GPU on an 8-GPU node);
          "stats"  nobody: the chunks stay where they were produced (finish() returns the local views) and only what
                   gather_episode_stats() carries -- a few KB per rollout -- crosses xGMI."""
        if mode not in self.MODES:
            raise ValueError("mode must be one of %r" % (self.MODES,))
        self.group = group
        self.mode, self.dst = mode, int(dst)
        self.pending = []   # (name, buffer, work)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.local_only = mode == "stats" or (self.world == 1 and not (always_collective and dist.is_initialized()))
        self._reserved = {}  # id(local tensor) -> receive buffer

    @property
    def receives(self):
        """does THIS rank end up holding the other ranks' chunks?"""
        return not self.local_only and (self.mode == "all" or self.rank == self.dst)

    def reserve(self, chunks):
        """Allocate the receive buffers of the given chunks (list of dicts name -> tensor) ahead of the rollout and run one
        tiny collective, so that neither hipMalloc nor RCCL's lazy channel setup lands between two step launches."""
        if self.local_only:
            return
        if self.receives:
            for local in chunks:
                for k in sorted(local):
                    v = local[k]
                    self._reserved[id(v)] = torch.empty((self.world,) + tuple(v.shape), dtype=v.dtype, device=v.device)
        v = next(iter(chunks[0].values()))
        warm_in = torch.zeros(64, dtype=torch.uint8, device=v.device)
        warm_out = torch.empty(64 * self.world, dtype=torch.uint8, device=v.device)
        if self.mode == "all":
            dist.all_gather_into_tensor(warm_out, warm_in, group=self.group)
        else:   # the same point-to-point channels the chunks will use
            dist.gather(warm_in, list(warm_out.view(self.world, 64).unbind(0)) if self.rank == self.dst else None,
                        dst=self._global_dst(), group=self.group)

    def _global_dst(self):
        return dist.get_global_rank(self.group, self.dst) if self.group is not None else self.dst

    def submit(self, local):
        """local: dict name -> tensor (a finished chunk; must not be written again)."""
        out = {}
        for k in sorted(local):
            v = local[k].contiguous()
            if self.local_only:
                out[k] = v.unsqueeze(0)
                self.pending.append((k, out[k], None))
                continue
            buf = self._reserved.pop(id(local[k]), None)
            if buf is None and self.receives:
                buf = torch.empty((self.world,) + tuple(v.shape), dtype=v.dtype, device=v.device)
            if self.mode == "all":
                work = dist.all_gather_into_tensor(buf.view(-1), v.view(-1), group=self.group, async_op=True)
            else:   # "root": rank dst receives world shards (its own included), everybody else only sends
                work = dist.gather(v, list(buf.unbind(0)) if self.receives else None, dst=self._global_dst(), group=self.group, async_op=True)
                if buf is None:
                    buf = v   # (kept alive until the send has completed)
            self.pending.append((k, buf, work))
            out[k] = buf if self.receives else None
        return out

    def finish(self):
        """Waits for every outstanding exchange; returns dict name -> list of [world, chunk, ...] tensors on a rank that receives
        (every rank in mode "all", rank dst in mode "root"), dict name -> [] on a rank that only sent, and the local chunks as
        [1, chunk, ...] views in mode "stats" / without a process group."""
        res = {}
        for k, buf, work in self.pending:
            if work is not None:
                work.wait()
            lst = res.setdefault(k, [])
            if self.local_only or self.receives:
                lst.append(buf)
        self.pending = []
        return res


def gather_episode_stats(returns, lengths, group=None):
    """Per-episode returns/lengths (KBs) to every rank: float32 [n_local_episodes, A], int32 [n_local_episodes]; the
    number of finished episodes differs per rank, so the result is dict name -> list of per-rank tensors."""
    return gather_ragged(dict(returns=returns, lengths=lengths), group=group)
