"""Waterworld C3 with StandardizedEnv(enable_obsnorm, enable_rewnorm): the wrapper fused into the step kernel vs the stand-alone
epilogue launches.  Prints time per wrapped step and the algorithmic HBM bytes per env-step of both."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrl_amd.waterworld import BatchedMAWaterWorld
from madrl_amd.wrappers import StandardizedEnv
dev = torch.device("cuda:0"); N = int(os.environ.get("MADRL_N", "32768"))
acts = [(torch.rand((N, 5, 2), device=dev) * 2 - 1).contiguous() for _ in range(8)]
for name, fused in (("epilogue kernels", False), ("fused", None)):
    env = StandardizedEnv(BatchedMAWaterWorld(5, 10, n_envs=N, device=dev, seed=0, auto_reset=True), scale_reward=0.5, enable_obsnorm=True,
                          enable_rewnorm=True, fused=fused)
    env.reset()
    for i in range(10): env.step(acts[i % 8])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(100): env.step(acts[i % 8])
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 100
    D = env.unwrapped.obs_dim; n_el = 5 * D
    sim = 40 + 20 + 1 + 8 + 2 * 416                       # actions, raw rewards, done, info, state record in + out
    b = sim + (36 * n_el + 36 * 5 if env._fused else 4 * n_el + 40 * n_el + 40 * 5)   # obs: raw store + read back vs none; float64 mean / var in + out, float32 out
    print("%-16s %.1f us per wrapped step   %.3e env-steps/s   %d algorithmic bytes per env-step -> %.0f GB/s" % (name, ms * 1e3, N / ms * 1e3, b, b * N / ms / 1e6))
