"""Policy-in-the-loop rollout throughput: Pursuit C2 (65 536 envs) with the device chase policy through RolloutCollector,
eager launches vs one hipGraph per horizon; plus a small batch where launch overhead dominates."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrl_amd.maps import rectangle_map
from madrl_amd.pursuit import BatchedPursuitEvade
from madrl_amd.heuristics import PursuitHeuristicPolicy
from madrl_amd.rollout import RolloutCollector
dev = "cuda:0"
ONLY = os.environ.get("ROLLOUT_ONLY", "")   # "single": the one-batch eager collector at 65 536 envs and nothing else (profiling runs)
for N, T in ((65536, 50),) if ONLY == "single" else ((65536, 50), (1024, 50)):
    for graph in (False,) if ONLY == "single" else (False, True):
        env = BatchedPursuitEvade([rectangle_map(16, 16)], n_envs=N, device=dev, seed=0, max_steps=500, auto_reset=True, n_pursuers=8,
                                  n_evaders=30, obs_range=7, n_catch=2, surround=True, flatten=True, reward_mech="local")
        col = RolloutCollector(env, PursuitHeuristicPolicy(7, flatten=True, seed=1), T, discount=0.99, graph=graph)
        for _ in range(3): col.collect()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        K = 6
        for _ in range(K): col.collect()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
        print("N=%6d T=%d graph=%-5s  %.2f ms per horizon  %.1f us per step  %.3e env-steps/s" % (N, T, graph, dt * 1e3, dt / T * 1e6, N * T / dt), flush=True)

# the same 65 536 envs as two sub-batches on their own HIP streams: one sub-batch's policy launch runs under the other's step kernel
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from madrl_amd.sharded import StreamSharded
from madrl_amd.rollout import ShardedRolloutCollector
# (MAX_BLOCKS: workgroups per sub-batch step launch, 0 = the resident capacity -- fewer leave wave slots for the other sub-batch's policy launch)
for S, graph, MB in () if ONLY == "single" else ((2, False, 0), (2, True, 0), (2, True, 4096), (2, True, 3072), (2, True, 2560), (4, False, 0), (4, True, 0), (4, True, 1280)):
    N, T = 65536, 50
    sh = StreamSharded(lambda n_envs, env_id_base, device: BatchedPursuitEvade([rectangle_map(16, 16)], n_envs=n_envs, device=device, seed=0, env_id_base=env_id_base,
                                                                                max_steps=500, auto_reset=True, n_pursuers=8, n_evaders=30, obs_range=7, n_catch=2,
                                                                                surround=True, flatten=True, reward_mech="local", max_blocks=MB), N, n_streams=S, device=dev)
    col = ShardedRolloutCollector(sh, [PursuitHeuristicPolicy(7, flatten=True, seed=1, row_id_base=j * (N // S) * 8) for j in range(S)], T, discount=0.99, graph=graph)
    for _ in range(3): col.collect()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 6
    for _ in range(K): col.collect()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    print("N=%6d T=%d  %d sub-batches on streams, graph=%-5s workgroups %-5s %.2f ms per horizon  %.1f us per step  %.3e env-steps/s" % (N, T, S, graph, MB or "dflt", dt * 1e3, dt / T * 1e6, N * T / dt), flush=True)
