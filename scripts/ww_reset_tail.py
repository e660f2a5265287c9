"""Waterworld C3: what the ~33 horizon resets that a steady-state launch carries cost, by launch shape (workgroups striding over the envs).
Episode ages uniform over [0, 1000) (33 resets per launch) against all ages 0 (none in the timed region)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrl_amd.waterworld import BatchedMAWaterWorld
from madrl_amd import _lib
dev = torch.device("cuda:0"); N = 32768
L = _lib.lib()
for blocks in (0, 32768, 8192, 6144):
    for spread in (False, True):
        env = BatchedMAWaterWorld(5, 10, n_envs=N, device=dev, seed=0, auto_reset=True, max_blocks=blocks)
        acts = [(torch.rand((N, 5, 2), device=dev) * 2 - 1).contiguous() for _ in range(8)]
        outs = [_lib.ptr(t) for t in (env._obs, env._rew, env._done, env._info)]
        env.reset()
        if spread:
            env.set_state(t=(torch.arange(N, device=dev, dtype=torch.int32) * 7919) % 1000)
        step = lambda i: L.madrl_waterworld_step(env._handle, _lib.ptr(acts[i % 8]), None, *outs, _lib.current_stream(dev))
        for i in range(20): step(i)
        torch.cuda.synchronize()
        ts = []
        for r in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(100): step(i)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 10)
        print("workgroups %5s  %s: min %.1f median %.1f us per step" % (blocks or "dflt", "ages spread (~33 resets per launch)" if spread else "ages 0 (no resets)              ", min(ts), sorted(ts)[2]), flush=True)
        del env
