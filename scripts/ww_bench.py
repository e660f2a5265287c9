"""Waterworld C3 timing (32 768 envs, 5 pursuers / 10 evaders / 10 poison / 30 sensors)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from madrl_amd.waterworld import BatchedMAWaterWorld
from madrl_amd import _lib
dev = torch.device("cuda:0"); N, Np = 32768, 5
env = BatchedMAWaterWorld(5, 10, n_envs=N, device=dev, seed=0, auto_reset=True)
acts = [(torch.rand((N, Np, 2), device=dev) * 2 - 1).contiguous() for _ in range(8)]
L = _lib.lib(); h = env._handle
ptrs = [_lib.ptr(t) for t in (env._obs, env._rew, env._done, env._info)]
env.reset()
def run(K):
    for i in range(K):
        _lib.check(L.madrl_waterworld_step(h, _lib.ptr(acts[i % 8]), None, *ptrs, _lib.current_stream(dev)))
B = 40 + 4260 + 20 + 1 + 8 + 2 * 416
for blocks in (2048, 4096, 6144, 8192, 16384, 32768):
    env.set_launch(blocks)
    run(20); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(100); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 100
    print("waterworld blocks=%6d  %.1f us/step  %.3e env-steps/s  %.0f GB/s (%d B/env-step)" % (blocks, ms * 1e3, N / ms * 1e3, B * N / ms / 1e6, B))
if "--cpu" in sys.argv:
    from oracle import waterworld as ww
    o = ww.WaterworldOracle(5, 10, n_envs=4096, seed=0, dtype=np.float32)
    o.reset(); a = np.random.uniform(-1, 1, (4096, 5, 2)).astype(np.float32)
    t0 = time.time(); k = 0
    while time.time() - t0 < 8: o.step(a); k += 1
    print("cpu oracle f32 (OpenMP): %.3e env-steps/s" % (4096 * k / (time.time() - t0)))
