import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from madrl_amd.multiwalker import BatchedMultiWalkerEnv
from oracle import multiwalker as mwo
N, T, W = 96, 40, 3
env = BatchedMultiWalkerEnv(n_walkers=W, position_noise=0, angle_noise=0, n_envs=N, device="cuda:0", seed=11, env_id_base=7)
orc = mwo.MultiWalkerOracle(n_walkers=W, position_noise=0.0, angle_noise=0.0, n_envs=N, seed=11, env_id_base=7)
obs = env.reset(); oobs = orc.reset()
print("reset obs err", np.abs(obs.cpu().numpy() - oobs).max())
rng = np.random.RandomState(3)
for t in range(T):
    w = np.zeros((N, env.world_bytes), np.uint8); w[:, :orc.world_bytes] = orc.worlds()
    env.state_buffer.copy_(torch.as_tensor(w, device="cuda:0"))
    act = rng.uniform(-1, 1, (N, W, 4)).astype(np.float32)
    obs, rew, done, _ = env.step(act); oobs, orew, odone = orc.step(act)
    b, f, _ = env.bodies(); ob, of = orc.bodies()
    e_obs = np.abs(obs.cpu().numpy() - oobs).reshape(N, -1).max(1)
    e_bod = np.abs(b.cpu().numpy() - ob).reshape(N, -1)
    e_pos = e_bod.reshape(N, -1, 6)[:, :, :3].reshape(N, -1).max(1); e_vel = e_bod.reshape(N, -1, 6)[:, :, 3:].reshape(N, -1).max(1)
    q = lambda a: "med %.1e p90 %.1e max %.1e" % (np.median(a), np.percentile(a, 90), a.max())
    print(t, "obs", q(e_obs), "| pos", q(e_pos), "| vel", q(e_vel), "| flags differ", int((f.cpu().numpy() != of).any(1).sum()), "done differ", int((done.cpu().numpy() != odone.astype(bool)).sum()))
    if odone.any(): orc.reset(mask=odone)
