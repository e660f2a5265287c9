"""why bench.py's MultiWalker line and scripts/mw_steady.py disagree: the bench loop with knobs"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrl_amd import _lib
from madrl_amd.multiwalker import BatchedMultiWalkerEnv
dev = torch.device("cuda:0"); N = 16384
L = _lib.lib()
for sync_every, use_ndone, nacts in ((0, True, 8), (0, False, 8)):
    env = BatchedMultiWalkerEnv(n_walkers=3, n_envs=N, device=dev, seed=0, auto_reset=True, max_steps=500)
    torch.manual_seed(0)
    acts = [(torch.rand((N, 3, 4), device=dev) * 2 - 1).contiguous() for _ in range(nacts)]
    outs = [_lib.ptr(t) for t in (env._obs, env._rew, env._done)]
    ndone = torch.zeros((), dtype=torch.int64, device=dev)
    env.reset()
    def step(i, rec):
        _lib.check(L.madrl_multiwalker_step(env._handle, _lib.ptr(acts[i % nacts]), *outs, _lib.current_stream(dev)))
        if rec and use_ndone: ndone.add_((env._done != 0).sum())
        if sync_every and i % sync_every == sync_every - 1: torch.cuda.synchronize()
    for i in range(200): step(i, False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(50): step(i, True)
    e1.record(); torch.cuda.synchronize()
    print("sync_every=%d ndone=%d nacts=%d: %.2f ms/step" % (sync_every, use_ndone, nacts, e0.elapsed_time(e1) / 50), flush=True)
    del env
