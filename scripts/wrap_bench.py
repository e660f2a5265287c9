"""Throughput of the epilogue kernels (wrappers, returns/GAE scan, heuristic policies) against the HBM roofline, at the
Pursuit C2 batch (65 536 envs x 8 agents x 148 floats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrl_amd import _lib
from madrl_amd.heuristics import PursuitHeuristicPolicy, WaterworldHeuristicPolicy, MultiWalkerHeuristicPolicy
dev = torch.device("cuda:0")
L = _lib.lib(); st = _lib.current_stream(dev); P = _lib.ptr
N, A, D, K = 65536, 8, 148, 4


def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def report(name, us, nbytes):
    print("%-34s %8.1f us  %7.0f GB/s algorithmic  (%.2f of 8 TB/s)" % (name, us, nbytes / us / 1e3, nbytes / us / 1e3 / 8000), flush=True)


obs = torch.rand(N, A, D, device=dev); out = torch.empty_like(obs)
mean = torch.zeros(N, A, D, dtype=torch.float64, device=dev); var = torch.ones_like(mean)
n = obs.numel()
report("obsnorm (f32 in/out, f64 stats rw)", timeit(lambda: _lib.check(L.madrl_wrap_obsnorm(P(obs), P(mean), P(var), P(out), n, n // N, None, 0.001, 1e-8, st))), n * (4 + 4 + 32))
buf = torch.zeros(N, A, D, K, device=dev)
report("obsbuffer k=4 (shift + append)", timeit(lambda: _lib.check(L.madrl_wrap_obsbuffer(P(obs), P(buf), n, n // N, K, None, None, st))), n * (4 + 2 * 4 * K))
rew = torch.rand(N, A, device=dev); rout = torch.empty_like(rew); rm = torch.zeros(N, A, dtype=torch.float64, device=dev); rv = torch.ones_like(rm)
report("rewnorm", timeit(lambda: _lib.check(L.madrl_wrap_rewnorm(P(rew), P(rm), P(rv), P(rout), N * A, A, None, 0.001, 1e-8, 1.0, 1, st))), N * A * 40)
T = 100
R = torch.rand(T, N, A, device=dev); Dn = (torch.rand(T, N, device=dev) < 0.01).to(torch.uint8); V = torch.rand(T + 1, N, A, device=dev)
ret = torch.empty_like(R); adv = torch.empty_like(R)
report("returns/GAE scan T=100", timeit(lambda: _lib.check(L.madrl_rollout_gae(P(R), P(Dn), P(V), T, N, A, 0.99, 0.95, P(ret), P(adv), st)), 20), T * N * (A * 16 + 1))
pol = PursuitHeuristicPolicy(7, flatten=True)
obs[..., 98:147] = (torch.rand(N, A, 49, device=dev) < 0.1).float() * 0.1
report("pursuit heuristic (reads ch2 only)", timeit(lambda: pol(obs)), N * A * (49 * 4 + 4))
wobs = torch.rand(32768, 5, 213, device=dev); wp = WaterworldHeuristicPolicy()
report("waterworld heuristic", timeit(lambda: wp(wobs)), 32768 * 5 * (4 * 30 * 4 + 8 + 8))
mobs = torch.rand(16384, 3, 32, device=dev); mp = MultiWalkerHeuristicPolicy()
report("multiwalker heuristic", timeit(lambda: mp(mobs)), 16384 * 3 * (14 * 4 + 16))
