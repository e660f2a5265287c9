"""One-off soak (GPU box): the MultiWalker kernels against the CPU build of the same source beyond what tests/ runs every time -- every
walker count 1..10 (all three capacity classes), other seeds, longer, both launch forms; the WHOLE per-env world record compared in every
byte after every step, mask resets on both sides.  python scripts/mw_gpu_soak.py [n_envs steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from madrl_amd.multiwalker import BatchedMultiWalkerEnv
from oracle import multiwalker as mwo

N, T = (int(a) for a in (sys.argv[1:3] + ["384", "300"])[:2])
total = 0
for W in range(1, 11):
    for fused in (False, True):
        seed = 4000 + 10 * W + int(fused)
        mech = "global" if (W + int(fused)) % 2 else "local"
        env = BatchedMultiWalkerEnv(n_envs=N, device="cuda:0", n_walkers=W, reward_mech=mech, seed=seed, env_id_base=W, position_noise=0.0, angle_noise=0.0)
        env.set_mode(fused=fused)
        orc = mwo.MultiWalkerOracle(n_walkers=W, position_noise=0.0, angle_noise=0.0, reward_mech=mech, n_envs=N, seed=seed, env_id_base=W)
        assert np.array_equal(env.reset().cpu().numpy(), orc.reset())
        rng = np.random.RandomState(seed)
        t0 = time.time(); nd = 0
        for t in range(T):
            act = rng.uniform(-1, 1, (N, W, 4)).astype(np.float32)
            if (t // 35) % 4 == 3:
                act[:] = 0
            obs, rew, done, _ = env.step(act)
            oobs, orew, odone = orc.step(act)
            assert np.array_equal(done.cpu().numpy(), odone.astype(bool)), (W, fused, t, "done")
            assert np.array_equal(obs.cpu().numpy(), oobs) and np.array_equal(rew.cpu().numpy(), orew), (W, fused, t, "obs / rewards")
            same = (env.state_buffer.cpu().numpy()[:, :orc.world_bytes] == orc.worlds()).all(axis=1)
            assert same.all(), (W, fused, t, "%d world records differ" % int((~same).sum()))
            nd += int(odone.sum())
            if odone.any():
                orc.reset(mask=odone); env.reset(mask=odone)
        assert not orc.overflow().any()
        total += N * T
        print("W=%2d %-12s %s reward: %d env-steps, every byte of every world record equal, %d episodes, %.0f s" % (W, "one launch" if fused else "three launches", mech, N * T, nd, time.time() - t0), flush=True)
print("mw gpu soak: clean, %d env-steps" % total)
