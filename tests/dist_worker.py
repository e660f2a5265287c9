"""Worker for tests/test_dist_cpu.py: world_size-2 gloo run of the multi-GPU layer on CPU tensors."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from madrl_amd.dist import shard_range, gather_trajectories, gather_episode_stats
    n_total = 1001
    lo, hi = shard_range(n_total)
    sizes = torch.zeros(world, dtype=torch.int64)
    sizes[rank] = hi - lo
    dist.all_reduce(sizes)
    assert int(sizes.sum()) == n_total and int(sizes.max() - sizes.min()) <= 1
    los = [shard_range(n_total, r, world)[0] for r in range(world)] + [n_total]
    assert all(shard_range(n_total, r, world)[1] == los[r + 1] for r in range(world))  # contiguous, disjoint
    # compact trajectory gather: every rank ends up with every rank's buffers, in rank order
    T, N, P = 7, 16, 3
    g = torch.Generator().manual_seed(100 + rank)
    local = dict(actions=torch.randint(0, 5, (T, N, P), generator=g, dtype=torch.uint8),
                 rewards=torch.randn((T, N, P), generator=g),
                 dones=torch.randint(0, 2, (T, N), generator=g, dtype=torch.uint8))
    out = gather_trajectories(local)
    for r in range(world):
        gr = torch.Generator().manual_seed(100 + r)
        exp = dict(actions=torch.randint(0, 5, (T, N, P), generator=gr, dtype=torch.uint8),
                   rewards=torch.randn((T, N, P), generator=gr),
                   dones=torch.randint(0, 2, (T, N), generator=gr, dtype=torch.uint8))
        for k in exp:
            assert out[k].shape == (world,) + tuple(exp[k].shape)
            assert torch.equal(out[k][r], exp[k]), (k, r)
    from madrl_amd.dist import ChunkedTrajectoryGather
    cg = ChunkedTrajectoryGather()
    chunks = []
    for c in range(3):
        gc = torch.Generator().manual_seed(1000 * rank + c)
        ch = dict(rewards=torch.randn((5, N, P), generator=gc), dones=torch.randint(0, 2, (5, N), generator=gc, dtype=torch.uint8))
        chunks.append(ch)
        cg.submit(ch)
    got = cg.finish()
    assert len(got["rewards"]) == 3 and got["rewards"][0].shape == (world, 5, N, P)
    for c in range(3):
        for r in range(world):
            gr = torch.Generator().manual_seed(1000 * r + c)
            exp_r = torch.randn((5, N, P), generator=gr)
            exp_d = torch.randint(0, 2, (5, N), generator=gr, dtype=torch.uint8)
            assert torch.equal(got["rewards"][c][r], exp_r) and torch.equal(got["dones"][c][r], exp_d), (c, r)
    # mode "root" (bench.py's default): rank 1 -- not 0, to see that dst is honoured -- receives every rank's chunks, the others only send;
    # mode "stats": nothing is exchanged, finish() hands back the local chunks
    for mode in ("root", "stats"):
        cg = ChunkedTrajectoryGather(mode=mode, dst=1)
        cg.reserve(chunks)
        for ch in chunks:
            cg.submit(ch)
        got = cg.finish()
        if mode == "stats":
            assert not cg.receives and all(torch.equal(got["rewards"][c][0], chunks[c]["rewards"]) for c in range(3))
        elif rank == 1:
            assert cg.receives and len(got["dones"]) == 3 and got["rewards"][0].shape == (world, 5, N, P)
            for c in range(3):
                for r in range(world):
                    gr = torch.Generator().manual_seed(1000 * r + c)
                    exp_r = torch.randn((5, N, P), generator=gr)
                    exp_d = torch.randint(0, 2, (5, N), generator=gr, dtype=torch.uint8)
                    assert torch.equal(got["rewards"][c][r], exp_r) and torch.equal(got["dones"][c][r], exp_d), (mode, c, r)
        else:
            assert not cg.receives and got["rewards"] == [] and got["dones"] == []
    # ragged: every rank finished a different number of episodes / owns an uneven env shard
    ne = 4 + 3 * rank
    st = gather_episode_stats(torch.full((ne, P), float(rank)), torch.full((ne,), rank, dtype=torch.int32))
    for r in range(world):
        assert st["returns"][r].shape == (4 + 3 * r, P) and bool((st["returns"][r] == float(r)).all())
        assert st["lengths"][r].shape == (4 + 3 * r,) and bool((st["lengths"][r] == r).all())
    from madrl_amd.dist import gather_ragged
    rg = gather_ragged(dict(rewards=torch.full((hi - lo, 2), float(rank))))  # shards of 1001 envs: 501 + 500
    assert [t.shape[0] for t in rg["rewards"]] == [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]
    assert all(bool((rg["rewards"][r] == float(r)).all()) for r in range(world))
    dist.barrier()
    dist.destroy_process_group()
    print("rank %d ok" % rank)


if __name__ == "__main__":
    main()
