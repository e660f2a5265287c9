"""Where a MultiWalker step's time goes, per wavefront (measurement build: SRC=multiwalker MACRO=MADRL_MW_TIMING EXTRA=-fno-slp-vectorize
scripts/variants.sh 1; run with MADRL_HIP_LIB=scripts/_variants/libmadrl_hip.multiwalker.1.so).  s_memtime stamps of every wavefront of the
solver launch and of the continuous-pass launch of ONE step in rollout steady state, with the per-env counters that explain them
(longest lane list, position iterations, continuous-pass events)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from madrl_amd import _lib
from madrl_amd.multiwalker import BatchedMultiWalkerEnv

dev = torch.device("cuda:0")
N, W = int(os.environ.get("MW_N", 16384)), 3
env = BatchedMultiWalkerEnv(n_walkers=W, n_envs=N, device=dev, seed=0, auto_reset=True, max_steps=500)
acts = [(torch.rand((N, W, 4), device=dev) * 2 - 1).contiguous() for _ in range(4)]
env.reset()
for i in range(int(os.environ.get("MW_WARM", 220))):
    env.step(acts[i % 4])
torch.cuda.synchronize()
L = _lib.lib()
L.madrl_multiwalker_debug_read.argtypes = [C.c_void_p, C.c_void_p]
MHZ = 100.0   # s_memtime counts shader clocks here (~2.1 GHz): the printed "us" are units of 100 clocks = 0.047 us
has_acc = hasattr(L, "madrl_multiwalker_debug_read_acc")
if has_acc:
    L.madrl_multiwalker_debug_read_acc.argtypes = [C.c_void_p, C.c_int]
for rep in range(3):
    if has_acc:
        acc = np.zeros((4096, 8), np.uint64)
        L.madrl_multiwalker_debug_read_acc(acc.ctypes.data_as(C.c_void_p), 1)   # zero the region accumulators
    env.step(acts[rep % 4])
    if has_acc:
        L.madrl_multiwalker_debug_read_acc(acc.ctypes.data_as(C.c_void_p), 0)
        nb_ = (N + 15) // 16
        a = acc[nb_:2 * nb_].astype(np.float64) / MHZ
        names = ("re-search after an event", "advance + update of the event's contact", "mini island (other contacts updated)", "20 position iterations",
                 "init + velocity sweeps", "integrate, SynchronizeFixtures, FindNewContacts, swept box", "first searches (pass 0)", "-")
        print("step %d, continuous pass, regions of the chains per wavefront (us, summed over the step): " % rep +
              " | ".join("%s: mean %.0f max %.0f" % (names[k], a[:, k].mean(), a[:, k].max()) for k in range(7)))
    stamps = np.zeros((2, 4096, 8), np.uint64)
    vals = np.zeros((2, 4096, 16, 4), np.int32)
    assert L.madrl_multiwalker_debug_read(stamps.ctypes.data_as(C.c_void_p), vals.ctypes.data_as(C.c_void_p)) == 0
    nb = (N + 15) // 16
    sb = nb   # the first nb blocks of a step launch rebuild spares; live envs follow
    for p, name, marks in ((0, "solve", ("start", "init", "warm", "velocity", "store", "position")), (1, "continuous pass", ("start", "post", "setup", "chains", "merge", "observe"))):
        st = stamps[p, sb:sb + nb].astype(np.float64)
        ok = st[:, 0] > 0
        last = 6 if p == 0 else 5
        seq = [0, 1, 2, 3, 4, 5, 6] if p == 0 else [0, 1, 2, 3, 4, 5]
        tot = (st[:, last] - st[:, 0]) / MHZ
        print("step %d, %s: %d wavefronts, total us: mean %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f" % (
            rep, name, ok.sum(), tot[ok].mean(), np.percentile(tot[ok], 50), np.percentile(tot[ok], 90), np.percentile(tot[ok], 99), tot[ok].max()))
        parts = []
        for a, b in zip(seq[:-1], seq[1:]):
            d = (st[:, b] - st[:, a]) / MHZ
            parts.append("%d->%d mean %.0f max %.0f" % (a, b, d[ok].mean(), d[ok].max()))
        print("    segments (us): " + " | ".join(parts))
        v = vals[p, sb:sb + nb]
        if p == 0:
            mc, nr, pi = v[..., 0].max(1), v[..., 1].max(1), v[..., 2].max(1)
            vel = (st[:, 3] - st[:, 2]) / MHZ
            pos = (st[:, 5] - st[:, 4]) / MHZ
            for k in range(0, 8):
                m = ok & (mc == k)
                if m.sum():
                    print("    longest lane list of the wavefront = %d: %4d wavefronts, velocity loop mean %.0f us, position loop mean %.0f us, total mean %.0f max %.0f" % (
                        k, m.sum(), vel[m].mean(), pos[m].mean(), tot[m].mean(), tot[m].max()))
            print("    rounds > 1 in %d wavefronts; position iterations of a wavefront: mean %.1f, ==60 in %.2f; of an env: mean %.1f, ==60 in %.2f, <=3 in %.2f" % (
                (nr > 1).sum(), pi[ok].mean(), (pi[ok] >= 60).mean(), v[..., 2][ok].mean(), (v[..., 2][ok] >= 60).mean(), (v[..., 2][ok] <= 3).mean()))
            print("    longest lane list of an ENV: histogram 0..7 %s" % np.bincount(v[..., 0][ok].ravel().clip(0, 7), minlength=8))
            print("    corr(total, longest list) %.2f  corr(total, position iterations) %.2f" % (np.corrcoef(tot[ok], mc[ok])[0, 1], np.corrcoef(tot[ok], pi[ok])[0, 1]))
        else:
            ne = v[..., 0]
            ch = (st[:, 3] - st[:, 2]) / MHZ
            if (st[ok][:, 6] > 0).all():   # two-pass continuous pass: stamp 6 sits between the first searches and the pending chains
                sr, evs = (st[:, 6] - st[:, 2]) / MHZ, (st[:, 3] - st[:, 6]) / MHZ
                print("    chains = first searches of every body: mean %.0f p90 %.0f max %.0f | pending chains (events): mean %.0f p90 %.0f max %.0f" % (
                    sr[ok].mean(), np.percentile(sr[ok], 90), sr[ok].max(), evs[ok].mean(), np.percentile(evs[ok], 90), evs[ok].max()))
            print("    events per env: mean %.2f; max over the wavefront's envs: histogram 0..8 %s" % (ne[ok].mean(), np.bincount(ne[ok].max(1).clip(0, 8), minlength=9)))
            for k in range(0, 7):
                m = ok & (ne.max(1) == k)
                if m.sum():
                    print("    most events of one env in the wavefront = %d: %4d wavefronts, chains mean %.0f us max %.0f" % (k, m.sum(), ch[m].mean(), ch[m].max()))
            print("    corr(chains time, sum of events in the wavefront) %.2f, corr(chains, max events) %.2f" % (np.corrcoef(ch[ok], ne[ok].sum(1))[0, 1], np.corrcoef(ch[ok], ne[ok].max(1))[0, 1]))
