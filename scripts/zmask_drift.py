"""Does the headline kernel's launch time drift as more observation cells become "known zero"?  One launch per step, 65 536 envs, windows
of 500 steps; masks starting at "nothing known" (what set_state / a new buffer give) and at "all zero" (a fresh buffer declared zero)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrl_amd.maps import rectangle_map
from madrl_amd.pursuit import BatchedPursuitEvade
from madrl_amd import _lib
dev = torch.device("cuda:0")
L = _lib.lib(); N, H = 65536, 500
for promise in (False, True):
    env = BatchedPursuitEvade([rectangle_map(16, 16)], n_envs=N, device=dev, seed=0, max_steps=H, auto_reset=True, n_pursuers=8, n_evaders=30,
                              obs_range=7, reward_mech="local", n_catch=2, surround=True, flatten=True)
    if not promise:
        env.invalidate_obs()
    env.reset()
    env.set_state(dict(t=(torch.arange(N, device=dev, dtype=torch.int32) * 7919) % H))
    acts = [torch.randint(0, 5, (N, 8), device=dev, dtype=torch.int32) for _ in range(8)]
    ptrs = [_lib.ptr(t) for t in (env._obs, env._rew, env._done, env._removed)]
    s = _lib.current_stream(dev)
    out, kinds = [], []
    import ctypes as C, numpy as np
    dbg = getattr(L, "madrl_pursuit_debug_slot_kinds", None) if hasattr(L, "madrl_pursuit_debug_slot_kinds") else None
    buf = np.zeros(8, np.uint64)
    if dbg: dbg(buf.ctypes.data_as(C.c_void_p), 1)
    for w in range(int(os.environ.get("WINDOWS", 12))):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(500):
            L.madrl_pursuit_step(env._handle, _lib.ptr(acts[i % 8]), None, *ptrs, s)
        torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / 500 * 1e6)
        if dbg:
            dbg(buf.ctypes.data_as(C.c_void_p), 1)
            kinds.append(" ".join("%.1f" % (v / 500.0 / N) for v in buf[:5]))
    if kinds:
        print("   (lane, slot) stores per env-step [inside | outside clean | outside dirty | partial clean | partial dirty] by window:\n      " + "\n      ".join(kinds))
    zoff = (env.record_bytes * N + 255) // 256 * 256
    zm = env._state[zoff:zoff + N * 256].view(torch.int32)
    unknown = float((zm != 0).float().mean())
    print("masks start at %-14s us per step by window of 500: %s | lanes with a flagged cell at the end: %.3f" % ("all zero:" if promise else "nothing known:", " ".join("%.1f" % v for v in out), unknown), flush=True)
    del env
