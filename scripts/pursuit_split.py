"""Headline shape (65 536 envs) as ONE launch per step against the same envs as S independent sub-batches, each on its own HIP stream.
A launch ends with a drain (the last wavefronts finish alone) and begins with every wavefront in lockstep; with sub-batches on their
own streams one sub-batch's drain overlaps another's ramp-up -- and, when the streams are not joined after every step, a sub-batch's
step t + 1 starts while another's step t is still draining.  Prints us per step of the WHOLE batch (wall clock over K steps)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from madrl_amd.maps import rectangle_map
from madrl_amd.pursuit import BatchedPursuitEvade
from madrl_amd import _lib
dev = torch.device("cuda:0")
P, N, K, H = 8, int(os.environ.get("MADRL_N", 65536)), 300, 500
L = _lib.lib()
kw = dict(n_pursuers=P, n_evaders=30, obs_range=7, n_catch=2, surround=True, flatten=True, reward_mech="local")


def make(n, base, max_blocks=0):
    env = BatchedPursuitEvade([rectangle_map(16, 16)], n_envs=n, device=dev, seed=0, env_id_base=base, max_steps=H, auto_reset=True, max_blocks=max_blocks, **kw)
    env.reset()
    env.set_state(dict(t=((torch.arange(n, device=dev, dtype=torch.int32) + base) * 7919) % H))
    acts = [torch.randint(0, 5, (n, P), device=dev, dtype=torch.int32) for _ in range(8)]
    ptrs = [_lib.ptr(t) for t in (env._obs, env._rew, env._done, env._removed)]
    return env, acts, ptrs


def timed(fn):
    fn(20); torch.cuda.synchronize()
    best = []
    for r in range(5):
        t0 = time.perf_counter(); fn(K); torch.cuda.synchronize(); best.append((time.perf_counter() - t0) / K * 1e6)
    return min(best), sorted(best)[2]


env, acts, ptrs = make(N, 0)
def single(k):
    s = _lib.current_stream(dev)
    for i in range(k):
        L.madrl_pursuit_step(env._handle, _lib.ptr(acts[i % 8]), None, *ptrs, s)
print("one launch per step:                          min %.1f  median %.1f us" % timed(single), flush=True)
del env
for S in (2, 4):
    for blocks in (0, 5120 // S):
        parts = [make(N // S, j * (N // S), blocks) for j in range(S)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
        torch.cuda.synchronize()

        def free(k):
            for i in range(k):
                for j, (e, a, p) in enumerate(parts):
                    L.madrl_pursuit_step(e._handle, _lib.ptr(a[i % 8]), None, *p, streams[j].cuda_stream)

        evs = [torch.cuda.Event() for _ in range(S)]
        main = torch.cuda.current_stream(dev)

        def joined(k):   # every step: fork from the caller's stream, join back into it
            for i in range(k):
                ev0 = torch.cuda.Event(); ev0.record(main)
                for j, (e, a, p) in enumerate(parts):
                    if j == 0:
                        L.madrl_pursuit_step(e._handle, _lib.ptr(a[i % 8]), None, *p, main.cuda_stream)
                    else:
                        streams[j].wait_event(ev0)
                        L.madrl_pursuit_step(e._handle, _lib.ptr(a[i % 8]), None, *p, streams[j].cuda_stream)
                        evs[j].record(streams[j]); main.wait_event(evs[j])
        import ctypes as C
        io = (_lib.PursuitShardIO * S)()
        hs = (C.c_void_p * S)(*[e._handle.value for e, _, _ in parts])
        for j, (e, a, p) in enumerate(parts):
            io[j].inj_evader_actions = None
            io[j].obs, io[j].rew, io[j].done, io[j].removed = (q.value for q in p)
            io[j].stream = streams[j].cuda_stream
        aptr = [[_lib.ptr(x).value for x in a] for _, a, _ in parts]

        def native(fork, join):
            def fn(k):   # ONE call of the C ABI per step: fork + S launches + join (madrl_pursuit_step_sharded)
                cs = main.cuda_stream
                for i in range(k):
                    for j in range(S):
                        io[j].actions = aptr[j][i % 8]
                    L.madrl_pursuit_step_sharded(hs, io, S, cs, fork, join)
            return fn
        print("%d sub-batches, %4s workgroups each, one C call per step, joined:       min %.1f  median %.1f us" % ((S, blocks or "dflt") + timed(native(1, 1))), flush=True)
        print("%d sub-batches, %4s workgroups each, one C call per step, free-running: min %.1f  median %.1f us" % ((S, blocks or "dflt") + timed(native(0, 0))), flush=True)
        print("%d sub-batches, %4s workgroups each, free-running streams:   min %.1f  median %.1f us" % ((S, blocks or "dflt") + timed(free)), flush=True)
        print("%d sub-batches, %4s workgroups each, joined after every step: min %.1f  median %.1f us" % ((S, blocks or "dflt") + timed(joined)), flush=True)
        if blocks == 0:   # the same fork / join as edges of ONE captured hipGraph of 50 steps (the runtime places the nodes)
            T = 50
            g = torch.cuda.CUDAGraph()
            cap = torch.cuda.Stream(device=dev)
            keep = []
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=cap):
                m = torch.cuda.current_stream(dev)
                for i in range(T):
                    ev0 = torch.cuda.Event(); ev0.record(m); keep.append(ev0)
                    for j, (e, a, p) in enumerate(parts):
                        if j == 0:
                            L.madrl_pursuit_step(e._handle, _lib.ptr(a[i % 8]), None, *p, m.cuda_stream)
                        else:
                            streams[j].wait_event(ev0)
                            L.madrl_pursuit_step(e._handle, _lib.ptr(a[i % 8]), None, *p, streams[j].cuda_stream)
                            ev = torch.cuda.Event(); ev.record(streams[j]); m.wait_event(ev); keep.append(ev)

            def graph_joined(k):
                for _ in range(max(1, k // T)):
                    g.replay()
            print("%d sub-batches, %4s workgroups each, joined after every step, one hipGraph per %d steps: min %.1f  median %.1f us" % ((S, "dflt", T) + timed(graph_joined)), flush=True)
            del g
        del parts
