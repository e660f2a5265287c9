"""MultiWalker C4 timing (16 384 envs, n_walkers = 3)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from madrl_amd.multiwalker import BatchedMultiWalkerEnv
from madrl_amd import _lib
dev = torch.device("cuda:0"); N, W = 16384, 3
env = BatchedMultiWalkerEnv(n_walkers=W, n_envs=N, device=dev, seed=0, auto_reset=True, max_steps=500)
acts = [(torch.rand((N, W, 4), device=dev) * 2 - 1).contiguous() for _ in range(4)]
env.reset()
for blocks in [int(a) for a in sys.argv[1:] if a.isdigit()] or (512, 1024, 2048, 4096):
    env.set_launch(blocks)
    env.reset()   # same simulation phase for every launch shape (the work per step grows as walkers fall)
    for i in range(3): env.step(acts[i % 4])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(10): env.step(acts[i % 4])
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("multiwalker blocks=%6d  %.2f ms/step  %.3e env-steps/s" % (blocks, ms, N / ms * 1e3))
if "--cpu" in sys.argv:
    from oracle import multiwalker as mwo
    o = mwo.MultiWalkerOracle(n_walkers=3, n_envs=1024, seed=0)
    o.reset(); a = np.random.uniform(-1, 1, (1024, 3, 4)).astype(np.float32)
    t0 = time.time(); k = 0
    while time.time() - t0 < 8:
        _, _, d = o.step(a); k += 1
        if d.any(): o.reset(mask=d)
    print("cpu build (OpenMP): %.3e env-steps/s" % (1024 * k / (time.time() - t0)))
