"""debug aid: which bytes of the world record differ between the HIP kernel and the CPU build of the same source"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from madrl_amd.multiwalker import BatchedMultiWalkerEnv
from oracle import multiwalker as mwo
N, W = 8, 3
env = BatchedMultiWalkerEnv(n_walkers=W, n_envs=N, device="cuda:0", seed=11, env_id_base=7, position_noise=0, angle_noise=0)
orc = mwo.MultiWalkerOracle(n_walkers=W, position_noise=0.0, angle_noise=0.0, n_envs=N, seed=11, env_id_base=7)
env.reset(); orc.reset()
rng = np.random.RandomState(3)
for t in range(6):
    g = env.state_buffer.cpu().numpy()[:, :orc.world_bytes]; c = orc.worlds()
    d = np.nonzero((g != c).any(0))[0]
    print("t=%d differing byte offsets (%d):" % (t, len(d)), d[:40], "world bytes", orc.world_bytes)
    if len(d):
        o = int(d[0]) // 4 * 4
        print(" first diff dword @%d: gpu %r cpu %r" % (o, g[0, o:o + 8].view(np.float32), c[0, o:o + 8].view(np.float32)), g[0, o:o+8], c[0, o:o+8])
    act = rng.uniform(-1, 1, (N, W, 4)).astype(np.float32)
    env.step(act); orc.step(act)
