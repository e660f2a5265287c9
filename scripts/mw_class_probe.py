"""Which (n_walkers, fused, auto_reset) combination of the MultiWalker launches fails on this box: each in its own process.
python scripts/mw_class_probe.py            (debugging aid; see tests/test_multiwalker_gpu.py for the real checks)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = """
import sys; sys.path.insert(0, %r)
import torch, numpy as np
from madrl_amd.multiwalker import BatchedMultiWalkerEnv
W, fused, ar, N, cont, steps = %d, %d, %d, %d, %d, %d
env = BatchedMultiWalkerEnv(n_walkers=W, n_envs=N, device='cuda:0', seed=1, auto_reset=bool(ar), position_noise=0.0, angle_noise=0.0, continuous_physics=bool(cont))
if fused: env.set_mode(fused=True)
o = env.reset(); torch.cuda.synchronize()
for t in range(steps):
    o, r, d, i = env.step(torch.zeros((N, W, 4), device='cuda:0'))
torch.cuda.synchronize()
print('ok', float(o.abs().sum()))
"""
for W, fused, ar, N, cont, steps in ((10, 1, 0, 4, 0, 0), (10, 1, 0, 4, 0, 5), (10, 1, 0, 4, 1, 0), (9, 0, 0, 4, 1, 5), (8, 1, 0, 4, 1, 5), (5, 1, 1, 40, 1, 5)):
    p = subprocess.run([sys.executable, "-c", CODE % (ROOT, W, fused, ar, N, cont, steps)], capture_output=True, text=True, env=dict(os.environ, AMD_LOG_LEVEL="1"))
    print("W=%d fused=%d auto_reset=%d N=%d continuous=%d steps=%d -> rc %d %s | %s" % (W, fused, ar, N, cont, steps, p.returncode, p.stdout.strip()[-60:], p.stderr.strip()[-300:].replace("\n", " / ")), flush=True)
