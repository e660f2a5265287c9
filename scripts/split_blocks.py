"""Two sub-batches of 32 768 envs on two streams (the headline configuration of bench.py): workgroups per launch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from madrl_amd.maps import rectangle_map
from madrl_amd.pursuit import BatchedPursuitEvade
from madrl_amd.sharded import shared_streams
from madrl_amd import _lib
dev = torch.device("cuda:0")
P, N, K, H = 8, 65536, 300, 500
L = _lib.lib()
kw = dict(n_pursuers=P, n_evaders=30, obs_range=7, n_catch=2, surround=True, flatten=True, reward_mech="local")
streams = shared_streams(dev, 2)
for blocks in (0, 3072, 4096, 4608, 5632, 6144, 8192, 10240):
    parts = []
    for j in range(2):
        n = N // 2
        env = BatchedPursuitEvade([rectangle_map(16, 16)], n_envs=n, device=dev, seed=0, env_id_base=j * n, max_steps=H, auto_reset=True, max_blocks=blocks, **kw)
        env.reset()
        env.set_state(dict(t=((torch.arange(n, device=dev, dtype=torch.int32) + j * n) * 7919) % H))
        acts = [torch.randint(0, 5, (n, P), device=dev, dtype=torch.int32) for _ in range(8)]
        parts.append((env, acts, [_lib.ptr(t) for t in (env._obs, env._rew, env._done, env._removed)]))
    def run(k):
        for i in range(k):
            for j, (e, a, p) in enumerate(parts):
                L.madrl_pursuit_step(e._handle, _lib.ptr(a[i % 8]), None, *p, streams[j].cuda_stream)
    run(20); torch.cuda.synchronize()
    ts = []
    for r in range(5):
        t0 = time.perf_counter(); run(K); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / K * 1e6)
    print("workgroups per launch %5s: min %.1f median %.1f us per step" % (blocks or "dflt", min(ts), sorted(ts)[2]), flush=True)
    del parts
