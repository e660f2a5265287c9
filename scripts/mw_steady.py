"""MultiWalker C4 timing by simulation phase (16 384 envs, n_walkers = 3; MW_N / MW_W change them): the work per step grows as walkers fall, so the
first steps after a reset are not the steady state of a rollout.  Prints ms / step in windows of 10 steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrl_amd.multiwalker import BatchedMultiWalkerEnv
dev = torch.device("cuda:0"); N, W = int(os.environ.get("MW_N", 16384)), int(os.environ.get("MW_W", 3))
import itertools
combos = [(True, True)] if '--one' in sys.argv else [(True, True), (False, True)] if '--quick' in sys.argv else list(itertools.product((True, False), (True, False)))
for cont, tof in combos:
    if True:
        env = BatchedMultiWalkerEnv(n_walkers=W, n_envs=N, device=dev, seed=0, auto_reset=True, max_steps=500, continuous_physics=cont,
                                    terminate_on_fall=tof)
        acts = [(torch.rand((N, W, 4), device=dev) * 2 - 1).contiguous() for _ in range(4)]
        env.reset()
        line = []
        for win in range(int(os.environ.get("MW_WINDOWS", 16))):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            nd = 0
            e0.record()
            for i in range(10):
                _, _, _, info = env.step(acts[i % 4])
            e1.record(); torch.cuda.synchronize()
            nd = int((info["done_bits"] != 0).sum())
            line.append("%.2f(%d)" % (e0.elapsed_time(e1) / 10, nd))
        print("continuous=%d terminate_on_fall=%d  ms/step per 10-step window (envs done in the last step): %s" % (cont, tof, " ".join(line)), flush=True)
        del env
