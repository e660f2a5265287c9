import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from madrl_amd.multiwalker import BatchedMultiWalkerEnv
from oracle import multiwalker as mwo
N, W = 8, 3
env = BatchedMultiWalkerEnv(n_walkers=W, position_noise=0, angle_noise=0, n_envs=N, device="cuda:0", seed=11, env_id_base=7)
orc = mwo.MultiWalkerOracle(n_walkers=W, position_noise=0.0, angle_noise=0.0, n_envs=N, seed=11, env_id_base=7)
obs = env.reset(); oobs = orc.reset()
gw = env.state_buffer.cpu().numpy()[:, :orc.world_bytes]; ow = orc.worlds()
diff = (gw != ow)
print("world bytes differing after reset:", int(diff.sum()), "of", diff.size, "first offsets", np.nonzero(diff.any(0))[0][:40])
b, f, _ = env.bodies(); ob, of = orc.bodies()
print("bodies err after reset", np.abs(b.cpu().numpy() - ob).max())
act = np.random.RandomState(3).uniform(-1, 1, (N, W, 4)).astype(np.float32)
obs, rew, done, _ = env.step(act); oobs, orew, odone = orc.step(act)
b, f, _ = env.bodies(); ob, of = orc.bodies()
print("own-state step: obs err", np.abs(obs.cpu().numpy() - oobs).max(), "bodies err", np.abs(b.cpu().numpy() - ob).max())
e = np.abs(b.cpu().numpy() - ob)[0]
print(np.round(e, 4))
gw = env.state_buffer.cpu().numpy()[:, :orc.world_bytes]; ow = orc.worlds()
n = 0
G = gw[n].copy(); O = ow[n].copy()
jt = np.dtype([("ix", "f4"), ("iy", "f4"), ("iz", "f4"), ("mi", "f4"), ("ms", "f4"), ("mt", "f4"), ("ls", "i4")])
st = np.dtype([("edge", "i2"), ("npts", "u1"), ("touch", "u1"), ("id", "u4", 2), ("ni", "f4", 2), ("ti", "f4", 2)])
gj, oj = G[504:504 + 448].view(jt), O[504:504 + 448].view(jt)
print("joints (gpu | cpu) ls, iz, mi:")
for k in range(12):
    print(k, gj[k]["ls"], oj[k]["ls"], "| iz %.4f %.4f | mi %.4f %.4f | ix %.3f %.3f iy %.3f %.3f" % (gj[k]["iz"], oj[k]["iz"], gj[k]["mi"], oj[k]["mi"], gj[k]["ix"], oj[k]["ix"], gj[k]["iy"], oj[k]["iy"]))
gs, os_ = G[952:952 + 166 * 28].view(st), O[952:952 + 166 * 28].view(st)
print("slots with edge>=0 (index, edge, npts, touch, ni):")
for k in range(166):
    if gs[k]["edge"] >= 0 or os_[k]["edge"] >= 0:
        if gs[k]["npts"] or os_[k]["npts"] or gs[k]["edge"] != os_[k]["edge"]:
            print(k, "gpu", gs[k]["edge"], gs[k]["npts"], gs[k]["touch"], np.round(gs[k]["ni"], 3), [hex(x) for x in gs[k]["id"]], "| cpu", os_[k]["edge"], os_[k]["npts"], os_[k]["touch"], np.round(os_[k]["ni"], 3), [hex(x) for x in os_[k]["id"]])
