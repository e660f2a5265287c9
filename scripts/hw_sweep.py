"""Hostage world step timing (32 768 envs) for a list of workgroup counts (profiling aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrl_amd.hostage import BatchedContinuousHostageWorld
from madrl_amd import _lib
dev = torch.device("cuda:0"); N, Nr = 32768, 3
env = BatchedContinuousHostageWorld(3, 10, 5, 2, 2, n_envs=N, device=dev, seed=0, auto_reset=True)
acts = [(torch.rand((N, Nr, 2), device=dev) * 2 - 1).contiguous() for _ in range(8)]
L = _lib.lib(); h = env._handle
ptrs = [_lib.ptr(t) for t in (env._obs, env._rew, env._done, env._info)]
env.reset()
def run(K):
    for i in range(K):
        _lib.check(L.madrl_hostage_step(h, _lib.ptr(acts[i % 8]), None, *ptrs, _lib.current_stream(dev)))
for blocks in [int(a) for a in sys.argv[1:]] or [4096, 8192, 16384]:
    env.set_launch(blocks)
    run(20); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(100); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 100
    print("hostage blocks=%6d  %.1f us/step  %.3e env-steps/s" % (blocks, ms * 1e3, N / ms * 1e3), flush=True)
