"""Workgroup-count sweep for the wave kernel (one wavefront per workgroup, persistent over envs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrl_amd.maps import rectangle_map
from madrl_amd.pursuit import BatchedPursuitEvade
from madrl_amd import _lib
dev = torch.device("cuda:0")
N, P = int(os.environ.get("MADRL_N", "65536")), 8
env = BatchedPursuitEvade([rectangle_map(16, 16)], n_envs=N, device=dev, seed=0, max_steps=500, auto_reset=True,
                          n_pursuers=P, n_evaders=30, obs_range=7, n_catch=2, surround=True, flatten=True, reward_mech="local")
acts = [torch.randint(0, 5, (N, P), device=dev, dtype=torch.int32) for _ in range(8)]
L = _lib.lib(); h = env._handle
ptrs = [_lib.ptr(t) for t in (env._obs, env._rew, env._done, env._removed)]
env.reset()
def run(K):
    for i in range(K):
        _lib.check(L.madrl_pursuit_step(h, _lib.ptr(acts[i % 8]), None, *ptrs, _lib.current_stream(dev)))
blocks_list = [int(a) for a in sys.argv[1:]] or [2048, 3072, 4096, 5120, 5461, 6144, 6554, 7168, 8192, 16384, 65536]
for blocks in blocks_list:
    env.set_launch(64, blocks)
    run(20); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(200); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 200
    print("blocks=%6d (%.2f envs/wave)  %.1f us/step  %.3e env-steps/s  %.0f GB/s" % (blocks, N / blocks, ms * 1e3, N / ms * 1e3, 5029 * N / ms / 1e6), flush=True)
