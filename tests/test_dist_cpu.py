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
