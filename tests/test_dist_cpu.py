"""CPU test of the N > 1 path: world_size 2, gloo backend, launched exactly like the driver
launches bench.py (torch.distributed.run on 127.0.0.1)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_single_process():
    from madrl_amd.dist import shard_range, gather_trajectories
    import torch
    assert shard_range(10, 0, 3) == (0, 4) and shard_range(10, 1, 3) == (4, 7) and shard_range(10, 2, 3) == (7, 10)
    assert shard_range(65536 * 8, 5, 8) == (5 * 65536, 6 * 65536)
    out = gather_trajectories(dict(r=torch.ones(2, 3)))  # no process group: world of one
    assert out["r"].shape == (1, 2, 3)


def test_gather_two_ranks_gloo():
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29731", os.path.join(ROOT, "tests", "dist_worker.py")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    assert "rank 0 ok" in p.stdout and "rank 1 ok" in p.stdout


def test_one_rank_group_still_issues_the_collectives():
    """ChunkedTrajectoryGather(always_collective=True) in a ONE-rank group goes through the backend's all-gather (what bench.py's
    MADRL_BENCH_FORCE_COLLECTIVE=1 uses to take the RCCL path on a one-GPU box) and returns what the local short cut returns."""
    code = r"""
import os, socket, torch, torch.distributed as dist
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
dist.init_process_group("gloo")
from madrl_amd.dist import ChunkedTrajectoryGather
chunks = [dict(actions=torch.arange(24, dtype=torch.uint8).view(2, 4, 3) + c, rewards=torch.randn(2, 4, 3), dones=torch.zeros(2, 4, dtype=torch.uint8)) for c in range(3)]
forced, local = ChunkedTrajectoryGather(always_collective=True), ChunkedTrajectoryGather()
assert not forced.local_only and local.local_only
forced.reserve(chunks)
assert len(forced._reserved) == 9
for c in chunks:
    forced.submit(c); local.submit(c)
a, b = forced.finish(), local.finish()
assert all(w is not None for w in [1]) and sorted(a) == sorted(b) == ["actions", "dones", "rewards"]
for k in a:
    assert len(a[k]) == len(b[k]) == 3
    for x, y in zip(a[k], b[k]):
        assert x.shape == y.shape and x.shape[0] == 1 and torch.equal(x, y) and x.data_ptr() != y.data_ptr()
dist.destroy_process_group()
print("one rank ok")
"""
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=dict(os.environ, OMP_NUM_THREADS="1"), capture_output=True, text=True, timeout=240)
    assert p.returncode == 0 and "one rank ok" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]
