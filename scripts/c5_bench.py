"""Pursuit C5 shard (32x32, 16 pursuers / 60 evaders, 32 768 envs = one GPU's share of BASELINE configs[4])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrl_amd.maps import rectangle_map
from madrl_amd.pursuit import BatchedPursuitEvade
from madrl_amd import _lib
dev = torch.device("cuda:0"); N, P, E = 32768, 16, 60
env = BatchedPursuitEvade([rectangle_map(32, 32)], n_envs=N, device=dev, seed=0, max_steps=500, auto_reset=True,
                          n_pursuers=P, n_evaders=E, obs_range=7, n_catch=2, surround=True, flatten=True, reward_mech="local")
print("kernel:", env.kernel_kind, "record bytes", env.record_bytes)
acts = [torch.randint(0, 5, (N, P), device=dev, dtype=torch.int32) for _ in range(8)]
L = _lib.lib(); h = env._handle
ptrs = [_lib.ptr(t) for t in (env._obs, env._rew, env._done, env._removed)]
env.reset()
B = 4 * P + 4 * P * 148 + 4 * P + 5 + 2 * (env.record_bytes)
for threads, blocks in ((64, 0), (64, 8192), (128, 0)):
    env.set_launch(threads, blocks)
    for i in range(10): _lib.check(L.madrl_pursuit_step(h, _lib.ptr(acts[i % 8]), None, *ptrs, _lib.current_stream(dev)))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(50): _lib.check(L.madrl_pursuit_step(h, _lib.ptr(acts[i % 8]), None, *ptrs, _lib.current_stream(dev)))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    print("C5 threads=%3d blocks=%5d  %.1f us/step  %.3e env-steps/s  %.0f GB/s (%d B/env-step)" % (threads, blocks, ms * 1e3, N / ms * 1e3, B * N / ms / 1e6, B))
