"""Step time of the authors' own training shapes (runners/old/rllab/pursuit.sh:1, runners/old/rltools/pursuit.sh:1: 32x32 map pool, 30 pursuers /
50 or 30 evaders, obs_range 11, --sample_maps --flatten --surround, local reward) on whatever kernel the library picks, and on the generic one:
    python scripts/authors_shape.py [n_envs] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from madrl_amd.maps import synthetic_map_pool
from madrl_amd.pursuit import BatchedPursuitEvade

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
pool = synthetic_map_pool(10, 32, 32)
for P, E in ((30, 50), (30, 30)):
    for kind in ("auto", "generic"):
        env = BatchedPursuitEvade(pool, n_envs=N, device="cuda:0", seed=0, max_steps=500, auto_reset=True, kernel=kind, n_pursuers=P, n_evaders=E, obs_range=11,
                                  sample_maps=True, flatten=True, surround=True, n_catch=2, reward_mech="local")
        env.reset()
        env.set_state(dict(t=(torch.arange(N, device="cuda:0", dtype=torch.int32) * 7919) % 500))
        acts = [torch.randint(0, 5, (N, P), dtype=torch.int32, device="cuda:0") for _ in range(8)]
        for i in range(30):
            env.step(acts[i % 8])
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            for i in range(K):
                env.step_into(acts[i % 8], env._rew, env._done)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / K)
        B = 4 * P + 4 * P * env.obs_dim + 4 * P + 5 + 2 * env.record_bytes
        print("%dv%d obs 11 32x32 pool, %d envs, kernel %-7s (%s): %.1f us/step  %.3e env-steps/s  %.0f GB/s = %.3f of 8 TB/s on %d B/env-step"
              % (P, E, N, kind, env.kernel_kind, best * 1e6, N / best, B * N / best / 1e9, B * N / best / 8e12, B), flush=True)
        del env
