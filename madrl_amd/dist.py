"""Multi-GPU layer: env instances never interact (each reference env object is self-contained;
the reference's own parallelism is N pickled copies handed to sampler workers,
runners/rurllab.py:259, runners/rurltools.py:184-191), so the path shards by env index with
NO data-path collective.  The single exchange is the episode-end gather of the compact
trajectory (actions, rewards, dones, per-episode info) -- the functional analogue of sampler
workers returning their paths -- done with one all_gather per tensor over RCCL/xGMI
(backend "nccl" on ROCm; "gloo" in the CPU tests).

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


def gather_trajectories(local, group=None):
    """local: dict name -> tensor with identical shape/dtype on every rank.
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


def gather_episode_stats(returns, lengths, group=None):
    """Per-episode returns/lengths (KBs) to every rank: float32 [n_local_episodes, A], int32 [n]."""
    return gather_trajectories(dict(returns=returns, lengths=lengths), group=group)
