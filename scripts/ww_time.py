"""Step time of the Waterworld kernel at BASELINE C3 (32 768 envs): python scripts/ww_time.py [n_envs] [max_blocks]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrl_amd.waterworld import BatchedMAWaterWorld
from madrl_amd import _lib
dev = torch.device("cuda:0")
for N in ([int(a) for a in sys.argv[1:]] or [32768]):
    env = BatchedMAWaterWorld(5, 10, n_envs=N, device=dev, seed=0, max_steps=500, auto_reset=True)
    if os.environ.get("MADRL_WW_BLOCKS"):
        env.set_launch(int(os.environ["MADRL_WW_BLOCKS"]))
    acts = [torch.rand(N, 5, 2, device=dev) * 2 - 1 for _ in range(8)]
    env.reset()
    L = _lib.lib()
    outs = [_lib.ptr(t) for t in (env._obs, env._rew, env._done, env._info)]
    def run(K):
        for i in range(K):
            _lib.check(L.madrl_waterworld_step(env._handle, _lib.ptr(acts[i % 8]), None, *outs, _lib.current_stream(dev)))
    run(30); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(300); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 300 * 1e3
    print("N=%6d  %.1f us/step  %.3e env-steps/s" % (N, us, N / us * 1e6), flush=True)
