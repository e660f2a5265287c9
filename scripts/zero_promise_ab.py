"""A/B in one process: the observation buffer declared all-zero at construction (the stale-zero masks start at "every cell known to be
zero") against the masks starting at "nothing known" (madrl_pursuit_invalidate_obs), headline shape and the secondary mode, one launch per
step; interleaved rounds."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrl_amd.maps import rectangle_map
from madrl_amd.pursuit import BatchedPursuitEvade
from madrl_amd import _lib
dev = torch.device("cuda:0")
L = _lib.lib(); N, K, H = 65536, 300, 500
for name, mode in (("headline (flatten, surround)", dict(n_catch=2, surround=True, flatten=True)), ("secondary (HWC, co-location)", dict(n_catch=4, surround=False, flatten=False))):
    res = {True: [], False: []}
    for rnd in range(3):
        for promise in (True, False):
            env = BatchedPursuitEvade([rectangle_map(16, 16)], n_envs=N, device=dev, seed=0, max_steps=H, auto_reset=True, n_pursuers=8, n_evaders=30,
                                      obs_range=7, reward_mech="local", **mode)
            if not promise:
                env.invalidate_obs()
            env.reset()
            env.set_state(dict(t=(torch.arange(N, device=dev, dtype=torch.int32) * 7919) % H))
            acts = [torch.randint(0, 5, (N, 8), device=dev, dtype=torch.int32) for _ in range(8)]
            ptrs = [_lib.ptr(t) for t in (env._obs, env._rew, env._done, env._removed)]
            s = _lib.current_stream(dev)
            def run(k):
                for i in range(k):
                    L.madrl_pursuit_step(env._handle, _lib.ptr(acts[i % 8]), None, *ptrs, s)
            run(50); torch.cuda.synchronize()
            t0 = time.perf_counter(); run(K); torch.cuda.synchronize()
            res[promise].append((time.perf_counter() - t0) / K * 1e6)
            del env
    print("%-30s promised zero: %s us | nothing known: %s us" % (name, " ".join("%.1f" % v for v in res[True]), " ".join("%.1f" % v for v in res[False])), flush=True)
