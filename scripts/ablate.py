"""Timing of the wave kernel under ablations / launch shapes (profiling aid)."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import torch
from madrl_amd.maps import rectangle_map
from madrl_amd.pursuit import BatchedPursuitEvade
from madrl_amd import _lib
dev = torch.device("cuda:0"); N, P = 65536, 8
env = BatchedPursuitEvade([rectangle_map(16, 16)], n_envs=N, device=dev, seed=0, max_steps=500, auto_reset=True,
                          n_pursuers=P, n_evaders=30, obs_range=7, n_catch=2, surround=True, flatten=True, reward_mech="local")
acts = [torch.randint(0, 5, (N, P), device=dev, dtype=torch.int32) for _ in range(8)]
L = _lib.lib(); h = env._handle
ptrs = [_lib.ptr(t) for t in (env._obs, env._rew, env._done, env._removed)]
env.reset()
def run(K):
    for i in range(K):
        _lib.check(L.madrl_pursuit_step(h, _lib.ptr(acts[i %% 8]), None, *ptrs, _lib.current_stream(dev)))
for blocks in [int(b) for b in sys.argv[1].split(",")]:
    env.set_launch(0, blocks)
    run(30); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(200); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 200
    print("ablate=%%s blocks=%%6d  %%.1f us/step  %%.3e env-steps/s  %%.0f GB/s" %% (os.environ.get("MADRL_PURSUIT_ABLATE", "0"), blocks, ms * 1e3, N / ms * 1e3, 5029 * N / ms / 1e6))
''' % ROOT
import sys as _s
SETS = (("0", "3072,4096,6144,8192"),)
for ab, blocks in SETS:
    env = dict(os.environ, MADRL_PURSUIT_ABLATE=ab)
    subprocess.run([sys.executable, "-c", CHILD, blocks], env=env)
