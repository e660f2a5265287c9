"""Launch time of the headline wave kernel against the number of envs (fixed 5 120 workgroups): t = a + b * N separates the
per-launch cost (dispatch ramp, LDS preamble, end-of-kernel write-back) from the per-env cost."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrl_amd.maps import rectangle_map
from madrl_amd.pursuit import BatchedPursuitEvade
from madrl_amd import _lib
dev = torch.device("cuda:0")
P = 8
sizes = [int(a) for a in sys.argv[1:]] or [5120, 10240, 20480, 30720, 40960, 51200, 61440, 65536, 71680]
rows = []
for N in sizes:
    env = BatchedPursuitEvade([rectangle_map(16, 16)], n_envs=N, device=dev, seed=0, max_steps=500, auto_reset=True,
                              n_pursuers=P, n_evaders=30, obs_range=7, n_catch=2, surround=True, flatten=True, reward_mech="local")
    acts = [torch.randint(0, 5, (N, P), device=dev, dtype=torch.int32) for _ in range(8)]
    L = _lib.lib(); h = env._handle
    ptrs = [_lib.ptr(t) for t in (env._obs, env._rew, env._done, env._removed)]
    env.reset()
    def run(K):
        for i in range(K):
            _lib.check(L.madrl_pursuit_step(h, _lib.ptr(acts[i % 8]), None, *ptrs, _lib.current_stream(dev)))
    run(20); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(200); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 200 * 1e3
    rows.append((N, us))
    print("N=%6d  %.1f us/step  %.3e env-steps/s" % (N, us, N / us * 1e6), flush=True)
    del env
import numpy as np
n, t = np.array(rows).T
b, a = np.polyfit(n, t, 1)
print("fit: t = %.1f us + %.3f ns * N" % (a, b * 1e3))
