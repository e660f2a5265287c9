"""Do HIP stream priorities change how two free-running sub-batches share the chip?  65 536 envs as two sub-batches; streams with equal priority
against one high- / one normal-priority stream; and against different workgroup counts per sub-batch.  us per step of the whole batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from madrl_amd.maps import rectangle_map
from madrl_amd.pursuit import BatchedPursuitEvade
from madrl_amd import _lib
dev = torch.device("cuda:0")
P, N, K, H = 8, 65536, 400, 500
L = _lib.lib()
kw = dict(n_pursuers=P, n_evaders=30, obs_range=7, n_catch=2, surround=True, flatten=True, reward_mech="local")
def make(n, base, max_blocks=0):
    env = BatchedPursuitEvade([rectangle_map(16, 16)], n_envs=n, device=dev, seed=0, env_id_base=base, max_steps=H, auto_reset=True, max_blocks=max_blocks, **kw)
    env.reset()
    env.set_state(dict(t=((torch.arange(n, device=dev, dtype=torch.int32) + base) * 7919) % H))
    acts = [torch.randint(0, 5, (n, P), device=dev, dtype=torch.int32) for _ in range(8)]
    ptrs = [_lib.ptr(t) for t in (env._obs, env._rew, env._done, env._removed)]
    return env, acts, ptrs
def timed(fn):
    fn(1500); torch.cuda.synchronize()
    best = []
    for r in range(5):
        t0 = time.perf_counter(); fn(K); torch.cuda.synchronize(); best.append((time.perf_counter() - t0) / K * 1e6)
    return min(best), sorted(best)[2]
CASES = [("equal priority", (0, 0), (0, 0)), ("high / normal priority", (-1, 0), (0, 0)), ("equal priority, 5120 / 2560 workgroups", (0, 0), (0, 2560))]
CASES += [("equal priority, %d workgroups each" % b, (0, 0), (b, b)) for b in (2560, 3072, 3328, 3584, 3840, 4096, 4352, 4608)]
if len(sys.argv) > 1:
    CASES = [("equal priority, %d workgroups each" % int(b), (0, 0), (int(b), int(b))) for b in sys.argv[1:]]
for name, prios, blocks in CASES:
    parts = [make(N // 2, j * (N // 2), blocks[j]) for j in range(2)]
    streams = [torch.cuda.Stream(device=dev, priority=p) for p in prios]
    torch.cuda.synchronize()
    def free(k):
        for i in range(k):
            for j, (e, a, p) in enumerate(parts):
                L.madrl_pursuit_step(e._handle, _lib.ptr(a[i % 8]), None, *p, streams[j].cuda_stream)
    print("%-45s min %.1f  median %.1f us per step" % ((name,) + timed(free)), flush=True)
    del parts
