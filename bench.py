#!/usr/bin/env python
"""bench.py -- env-steps/sec of the batched rollout engine (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python bench.py --gpus 8 --workload pursuit_c5        # BASELINE configs[4]: 32x32, 16 v 60, 262 144 envs over 8 GPUs

One "step" = one pass of the hot path over one batch: a single launch of the fused step kernel (pre-move reward, moves,
catch resolution, observations, fused auto-reset) over `--envs` env instances per GPU (default 65 536 = BASELINE
configs[1], PursuitEvade 16x16, 8 pursuers / 30 evaders, obs_range 7, surround).  Inputs (the pursuer action tensors) are
resident in HBM before the timed region starts; evader actions are drawn in-kernel (Philox).  The env instances start the
timed region at episode ages spread uniformly over [0, horizon), so EVERY launch carries its N / horizon share of fused
auto-resets (the two-observation-pass path), as in a steady rollout.

N > 1: one rank per GPU -- started by torch.distributed.run (the driver's command line), or by this script itself when it is
invoked as plain `python bench.py --gpus N` (it re-executes itself under torch.distributed.run on 127.0.0.1).  Env index
ranges are sharded by rank (weak scaling), the compact trajectory (actions, rewards, dones) of the timed region is
exchanged over RCCL in chunks that overlap with the stepping and waited for at the end, inside the timed region: `--gather root`
(default) gathers it on rank 0 -- north_star's "RCCL gather of trajectory buffers at episode end" -- `all` on every rank
(all-gather), `stats` exchanges per-episode returns only.  At N = 1 `workloads.pursuit_collective` runs that code path in a
one-rank group (the headline config with recording + exchange in the timed region; `over_headline` = its step time / the headline's).

Timing: W untimed warm-up steps, then REPEATS regions of EXACTLY K steps, each bracketed by barrier + synchronize on both
sides, max over ranks; `value` / `ms_per_step` are the MEDIAN region, `config.region_ms_per_step` lists all of them.

Rank 0 prints ONE JSON line (contract in the task description) with `roofline` and `cpu_baseline`; at N = 1 the same line
carries `workloads`: the other BASELINE configs (Waterworld configs[2], MultiWalker configs[3], the per-GPU shard of
configs[4]), the survey's secondary Pursuit mode and Waterworld under StandardizedEnv, timed the same way for a bounded
number of steps, each with its own roofline and its own bounded cpu_baseline.

cpu_baseline = the C restatement of the reference algorithm (oracle/, kind "port") timed LIVE on this box's host cores;
when MADRL_REFERENCE_ROOT names a checkout of the reference (never the case on the GPU box, where the tree does not
exist) the unmodified reference's NumPy path is timed live instead (kind "reference").  `cpu_reference_recorded` quotes the
committed record of that NumPy path taken in the build container (profiles/*_cpu_reference/record.json), host labelled.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# the sub-batch streams of a GPU (madrl_amd/sharded.py) must not share a hardware queue: ROCm multiplexes HIP streams onto
# GPU_MAX_HW_QUEUES queues (4 by default); read by the HIP runtime when it starts, so set before torch is imported
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# multi-process GPU work on this pool: the host driver only supports dmabuf IPC (without it RCCL fails with `hipIpcGetMemHandle: invalid argument`);
# already exported on the build and GPU boxes -- set here too so that a bare shell gets it, before the HIP runtime starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md)
HBM_COPY_GBPS = 6290.0  # the same guide's measured copy rate (MI355X_MICROARCH.md:34-35): what a pure streaming kernel reaches
VALU_PEAK_TFLOPS = 157.3  # FP32 vector peak, same guide
REPEATS = 3             # timed K-step regions per workload (each bracketed by barrier + synchronize); the line reports the median


def algorithmic_bytes_per_env_step(P, E, D, rec_bytes):
    """DESIGN.md "Algorithmic bytes": actions in, obs/rewards/done/removed out, packed state
    record read + written once."""
    return 4 * P + 4 * P * D + 4 * P + 1 + 4 + 2 * rec_bytes


def measured_traffic(envs_per_launch, workload="pursuit", launches_per_step=1):
    """HBM bytes per STEP (= per launch x the launches of a step) from the committed rocprofv3 PMC passes
    (profiles/*/pmc_traffic.json, FETCH_SIZE + WRITE_SIZE collected in separate runs by scripts/profile_workload.sh); None if absent
    or taken at another launch size.  bench.py cannot collect PMC counters on itself."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_traffic.json"))):
        try:
            j = json.load(open(f))
        except Exception:
            continue
        if int(j.get("envs", -1)) == int(envs_per_launch) and j.get("workload", "pursuit") == workload:
            best = (float(j["traffic_bytes_per_launch"]) * launches_per_step, os.path.relpath(f, ROOT))
    return best


def cpu_reference_record(key):
    """The UNMODIFIED reference's NumPy path as recorded by scripts/cpu_reference_bench.py in the build container (the
    reference tree cannot travel to the GPU box): quoted, host labelled, never re-timed here.  None without a record."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_cpu_reference", "record.json")), reverse=True):
        try:
            j = json.load(open(f))
            r = j[key]
        except Exception:
            continue
        return dict(value=r["all_core_steps_per_s"], unit="env-steps/s", cores=r["processes"], kind="reference-numpy (recorded)",
                    one_process_value=r["one_process_steps_per_s"],
                    host="%s, %d logical cores (%s)" % (j["host"]["cpu_model"], j["host"]["logical_cores"], j["host"]["where"]),
                    source=os.path.relpath(f, ROOT),
                    sample="%s; one env per process x %d processes x %d steps, OMP_NUM_THREADS=1; recorded, not re-timed by this run"
                           % (r["config"], r["processes"], r["steps_per_process"]))
    return None


def cpu_reference_live(which, budget_steps=1200):
    """The unmodified reference timed NOW on this host's cores -- only when MADRL_REFERENCE_ROOT is set and holds the tree
    (build container / a maintainer's checkout; never on the GPU box).  Runs scripts/cpu_reference_bench.py's workers in a
    child process (they pin OMP_NUM_THREADS=1 and fork a Pool over all cores)."""
    root = os.environ.get("MADRL_REFERENCE_ROOT")
    if not root or not os.path.isdir(os.path.join(root, "madrl_environments")):
        return None
    import subprocess
    code = ("import json, os, sys; sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'scripts'))\n"
            "import cpu_reference_bench as b\n"
            "r = b.measure(b.run_%s, %d, os.cpu_count()); r['cpu_model'] = b.cpu_model(); print(json.dumps(r))" % (ROOT, ROOT, which, budget_steps))
    try:
        out = subprocess.check_output([sys.executable, "-c", code], env=dict(os.environ, MPLBACKEND="Agg"), timeout=180).decode()
        r = json.loads(out.strip().splitlines()[-1])
    except Exception as e:  # a broken checkout must not take the bench line down
        return dict(error="reference timing failed: %r" % (e,))
    return dict(value=r["all_core_steps_per_s"], unit="env-steps/s", cores=r["processes"], kind="reference",
                one_process_value=r["one_process_steps_per_s"],
                sample="unmodified reference (%s) from MADRL_REFERENCE_ROOT, one env per process x %d processes x %d steps, "
                       "OMP_NUM_THREADS=1, timed live on %s" % (which, r["processes"], r["steps_per_process"], r["cpu_model"]))


def cpu_baseline_port(maps, kw, budget_s=10.0):
    """The C oracle (a port of the reference's algorithm) on the host cores, OpenMP over envs.
    Bounded sample of the same workload: 4096 envs, free-running, ~budget_s seconds."""
    import numpy as np
    from oracle import pursuit as po
    n = 4096
    orc = po.PursuitOracle(maps, n_envs=n, seed=0, **kw)
    orc.reset()
    rng = np.random.RandomState(0)
    acts = [rng.randint(5, size=(n, kw["n_pursuers"])).astype(np.int32) for _ in range(8)]
    orc.step(acts[0])
    t0 = time.time()
    steps = 0
    while time.time() - t0 < budget_s:
        _, _, done, _ = orc.step(acts[steps % 8])
        steps += 1
        if steps % 500 == 0:
            orc.reset()
    dt = time.time() - t0
    return dict(value=n * steps / dt, unit="env-steps/s", cores=po.lib().po_num_threads(), kind="port",
                sample="C oracle (oracle/pursuit_oracle.c, OpenMP), %d envs x %d steps, same config, %.1f s" % (n, steps, dt))


def cpu_baseline_rollout(budget_s=3.0):
    """The policy-in-the-loop workload on the host: the C oracle's step (OpenMP over envs) and the NumPy restatement of the reference's chase
    policy (heuristics/pursuit.py:18-54, one row at a time as the reference calls it) choosing every action from the observation the step
    just wrote.  Bounded sample: 256 envs, ~budget_s seconds."""
    import numpy as np
    from madrl_amd.maps import rectangle_map
    from oracle import pursuit as po
    from oracle.heuristics_oracle import pursuit_actions
    MS, P, E, _, mode = PURSUIT_VARIANTS["pursuit"]
    n, R = 256, 7
    orc = po.PursuitOracle([rectangle_map(MS, MS)], n_envs=n, seed=0, n_pursuers=P, n_evaders=E, obs_range=R, reward_mech="local", **mode)
    obs = orc.reset()
    rng = np.random.RandomState(0)
    t0 = time.time()
    steps = 0
    while time.time() - t0 < budget_s or steps < 2:
        win = np.asarray(obs).reshape(n * P, -1)[:, :3 * R * R].reshape(n * P, 3, R, R).transpose(0, 2, 3, 1)   # flattened rows: channel-major (:448-449)
        a = pursuit_actions(win)
        a = np.where(a < 0, rng.randint(5, size=a.shape), a).astype(np.int32).reshape(n, P)
        obs = orc.step(a)[0]
        steps += 1
    dt = time.time() - t0
    return dict(value=n * steps / dt, unit="env-steps/s", cores=po.lib().po_num_threads(), kind="port",
                sample="C oracle step (OpenMP) + NumPy chase policy row by row (oracle/heuristics_oracle.py, one Python thread), %d envs x %d steps, %.1f s" % (n, steps, dt))


def attach_cpu_baselines(out, live_ref_key, record_key, port_fn):
    """cpu_baseline is always a LIVE measurement on this host (the reference when a checkout is reachable through
    MADRL_REFERENCE_ROOT, else the C port); the committed record of the reference goes under its own key."""
    live = cpu_reference_live(live_ref_key) if live_ref_key else None
    port = port_fn()
    if live is not None and "error" not in live:
        out["cpu_baseline"], out["cpu_baseline_port"] = live, port
    else:
        out["cpu_baseline"] = port
        if live is not None:
            out["cpu_baseline_reference_error"] = live["error"]
    rec = cpu_reference_record(record_key) if record_key else None
    if rec is not None:
        out["cpu_reference_recorded"] = rec


def collective_on(world):
    """The N > 1 machinery (process group, barrier, max over ranks, the chunked trajectory all-gather inside the timed region) runs when
    there is more than one rank -- or when MADRL_BENCH_FORCE_COLLECTIVE=1 asks for it with ONE rank: the only way to take the RCCL
    ("nccl") code path on a one-GPU box (two ranks on one device are refused by RCCL; tests/test_bench_contract_gpu.py)."""
    return world > 1 or os.environ.get("MADRL_BENCH_FORCE_COLLECTIVE") == "1"


class Timer(object):
    """the bench contract: W untimed warm-up steps, then EXACTLY K steps bracketed by a barrier + synchronize on both sides;
    HIP events on the launch stream give the average launch duration"""

    def __init__(self, world, dev, streams=None, coll=None):
        self.world, self.dev, self.streams = world, dev, streams
        self.coll = collective_on(world) if coll is None else coll

    def barrier(self):
        import torch
        if self.coll:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def region(self, step, K, tail=None):
        import torch
        self.barrier()
        streams = self.streams or [torch.cuda.current_stream(self.dev)]
        ev0 = [torch.cuda.Event(enable_timing=True) for _ in streams]
        ev1 = [torch.cuda.Event(enable_timing=True) for _ in streams]
        t0 = time.perf_counter()
        for e, st in zip(ev0, streams):
            e.record(st)
        for i in range(K):
            step(i, True)
        for e, st in zip(ev1, streams):
            e.record(st)  # events bracket the K step launches of every stream (and the small trajectory copies when N > 1)
        if tail is not None:
            tail()
        self.barrier()
        dt = time.perf_counter() - t0
        # HIP events on the streams the kernels are launched on: duration of one step = the slowest stream's K launches / K.  With
        # several streams the launches of one step run CONCURRENTLY (one per stream), so this is the duration of the step, not of a
        # kernel running alone
        kernel_ms = max(a.elapsed_time(b) for a, b in zip(ev0, ev1)) / K
        if self.coll:
            import torch.distributed as dist
            tmax = torch.tensor([dt], dtype=torch.float64, device=self.dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt, kernel_ms

    def run(self, step, K, W, tail=None, prepare=None, repeats=REPEATS):
        """-> (dt, kernel_ms) of the MEDIAN region (by wall time) and the list of all regions' ms per step.  `prepare` runs
        before every region, outside it (receive buffers of the trajectory gather)."""
        for i in range(W):
            step(i, False)
        regions = []
        for _ in range(max(1, repeats)):
            if prepare is not None:
                prepare()
            regions.append(self.region(step, K, tail))
        order = sorted(range(len(regions)), key=lambda r: regions[r][0])
        dt, kernel_ms = regions[order[len(order) // 2]]
        return dt, kernel_ms, [r[0] / K * 1e3 for r in regions]


def region_stats(region_ms):
    s = sorted(region_ms)
    return {"timed_regions": len(s), "region_ms_per_step": [round(x, 6) for x in region_ms], "region_ms_per_step_min": s[0],
            "region_ms_per_step_median": s[len(s) // 2], "value_is": "median region"}


def roofline(bytes_per, N, kernel_ms, wall_ms, traffic, kernel, streams=1, **more):
    """N envs per step and GPU, stepped as `streams` concurrent launches of N / streams envs (one per HIP stream).  `achieved` / `frac` = the
    algorithmic bytes of the launches that run together / the duration of a step by the WALL CLOCK of the timed region (ms_per_step: what the
    driver's own clock sees); `achieved_kernel` / `frac_kernel` = the same bytes / the duration by HIP events on every launch stream, which
    leaves out the host's share of the region (2 - 3 % kinder)."""
    achieved = bytes_per * N / (wall_ms * 1e-3) / 1e9
    achieved_k = bytes_per * N / (kernel_ms * 1e-3) / 1e9
    out = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
           "achieved_kernel": achieved_k, "frac_kernel": achieved_k / HBM_PEAK_GBPS,
           "frac_vs_measured_copy": achieved / HBM_COPY_GBPS, "measured_copy_peak": HBM_COPY_GBPS,
           "traffic": traffic[0] if traffic else None, "traffic_source": traffic[1] if traffic else None,
           "algorithmic_bytes_per_launch": bytes_per * N // streams, "concurrent_launches_per_step": streams,
           "algorithmic_bytes_per_step": bytes_per * N, "kernel": kernel, "kernel_ms": kernel_ms,
           "kernel_ms_is": ("duration of one step = %d launches of %d envs each running CONCURRENTLY on %d HIP streams (events on every stream, "
                            "slowest stream); a launch's own wall time is about the same, its share of the chip is 1 / %d" % (streams, N // streams, streams, streams))
                           if streams > 1 else "average duration of the step's launch (HIP events on its stream)",
           "algorithmic_bytes_per_env_step": bytes_per}
    out.update(more)
    return out


# sub-batches per GPU, each on its own HIP stream, when --streams is not given: what scripts/stream_sweep.sh measured fastest on MI355X
# (the one-launch-per-step figure is reported next to it in every line: roofline.one_launch_per_step)
DEFAULT_STREAMS = {"pursuit": 2, "pursuit_c5": 2, "pursuit_colocate": 2, "pursuit_authors": 2, "waterworld": 2, "waterworld_std": 2, "hostage": 2, "multiwalker": 4, "multiwalker_w10": 4}


def shard_count(args, N, workload):
    S = max(1, int(args.streams)) if args.streams else DEFAULT_STREAMS[workload]
    return S if N % S == 0 and N // S >= 64 else 1


PURSUIT_VARIANTS = {
    # name: (map side, pursuers, evaders, default envs per GPU, env kwargs)
    "pursuit": (16, 8, 30, 65536, dict(n_catch=2, surround=True, flatten=True)),              # BASELINE configs[1]
    "pursuit_c5": (32, 16, 60, 32768, dict(n_catch=2, surround=True, flatten=True)),          # configs[4], one GPU's shard
    # SURVEY 8 preamble's secondary mode: heuristics/pursuit.py:66-67 (co-location catch, (R, R, 4) observations)
    "pursuit_colocate": (16, 8, 30, 65536, dict(n_catch=4, surround=False, flatten=False)),
    # the authors' own training shape (runners/old/rllab/pursuit.sh:1: 30 pursuers / 50 evaders, obs_range 11, --sample_maps --flatten --surround
    # on a 32x32 map pool; their map_pool32.npy is not in the tree -> a synthetic pool of ten maps, madrl_amd/maps.py)
    "pursuit_authors": (32, 30, 50, 16384, dict(n_catch=2, surround=True, flatten=True, obs_range=11, sample_maps=True, pool=10)),
}


def ensure_group(dev):
    """a process group for the trajectory exchange: the launcher's when there is one, else a ONE-rank group of this process (how a one-GPU box
    takes the RCCL code path: workloads.pursuit_collective, MADRL_BENCH_FORCE_COLLECTIVE=1); -> True when this call created it"""
    import torch.distributed as dist
    if dist.is_initialized():
        return False
    if "MASTER_ADDR" not in os.environ:
        import socket
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sk.getsockname()[1]), RANK="0", WORLD_SIZE="1")
        sk.close()
    backend = os.environ.get("MADRL_BENCH_BACKEND", "nccl")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)
    return True


def bench_pursuit(args, variant, K, W, rank, world, dev, cpu_budget, streams=None, reference_pass=True, api_leg=None, coll=None):
    import numpy as np
    import torch
    from madrl_amd.maps import rectangle_map
    from madrl_amd.pursuit import BatchedPursuitEvade
    from madrl_amd import _lib
    MS, P, E, N0, mode = PURSUIT_VARIANTS[variant]
    mode = dict(mode)
    N, R = (args.envs or N0), mode.pop("obs_range", 7)
    H = args.horizon
    S = shard_count(args, N, variant) if streams is None else streams
    per = N // S
    n_pool = mode.pop("pool", 0)
    if n_pool:
        from madrl_amd.maps import synthetic_map_pool
        maps = list(synthetic_map_pool(n_pool, MS, MS, seed=0))
    else:
        maps = [rectangle_map(MS, MS)]
    kw = dict(n_pursuers=P, n_evaders=E, obs_range=R, reward_mech="local", **mode)
    # the batch as S independent sub-batches, each on its own HIP stream (madrl_amd/sharded.py): env ids continue across them
    envs = [BatchedPursuitEvade(maps, n_envs=per, device=dev, seed=0, env_id_base=rank * N + j * per, max_steps=H,
                                auto_reset=True, threads=args.threads, max_blocks=args.max_blocks, **kw) for j in range(S)]
    from madrl_amd.sharded import shared_streams
    hip_streams = shared_streams(dev, S) if S > 1 else [torch.cuda.current_stream(dev)]
    D = envs[0].obs_dim
    rec_bytes = envs[0].record_bytes
    if S > 1 and algorithmic_bytes_per_env_step(P, E, D, rec_bytes) * N > 375e6:
        # the library walks the env range in alternating directions once ONE launch moves more than ~375 MB (the rows written last are
        # then touched first, while the 256 MB memory-side cache still holds them); here the sub-batches together are that large
        for e in envs:
            e.set_walk("alternate")
    gen = torch.Generator(device=dev).manual_seed(rank)
    n_act = 16
    actions = [torch.randint(0, 5, (N, P), generator=gen, device=dev, dtype=torch.int32) for _ in range(n_act)]
    # compact trajectory of the timed region (what a sampler returns to the learner), cut into
    # chunks whose all-gather over RCCL/xGMI overlaps with the stepping of the next chunk
    coll = collective_on(world) if coll is None else coll
    gather_mode = getattr(args, "gather", "root")
    CH = 50
    n_chunks = (K + CH - 1) // CH
    chunk_len = [min(CH, K - c * CH) for c in range(n_chunks)]
    traj = [dict(actions=torch.zeros((chunk_len[c], N, P), dtype=torch.uint8, device=dev),
                 rewards=torch.zeros((chunk_len[c], N, P), dtype=torch.float32, device=dev),
                 dones=torch.zeros((chunk_len[c], N), dtype=torch.uint8, device=dev)) for c in range(n_chunks)] if coll else []
    L = _lib.lib()
    hs = [e._handle for e in envs]
    outs = [[_lib.ptr(t) for t in (e._obs, e._rew, e._done, e._removed)] for e in envs]
    sp = [C_void(st.cuda_stream) for st in hip_streams]
    act_p = [[_lib.ptr(a[j * per:(j + 1) * per]) for a in actions] for j in range(S)]
    # N > 1: the step kernel writes rewards / dones straight into their slot of the trajectory chunk (the C ABI takes any
    # device pointer), only the action row is copied (int32 -> uint8)
    slot_p = [[(_lib.ptr(traj[c]["rewards"][q][j * per:(j + 1) * per]), _lib.ptr(traj[c]["dones"][q][j * per:(j + 1) * per]))
               for c in range(n_chunks) for q in range(chunk_len[c])] for j in range(S)] if coll else []
    gatherer = None
    if coll:
        from madrl_amd.dist import ChunkedTrajectoryGather
        gatherer = ChunkedTrajectoryGather(always_collective=True, mode=gather_mode)

    def prepare():
        if gatherer is not None:
            gatherer.reserve(traj)   # receive buffers + RCCL channel setup stay out of the timed region

    def submit(chunk):
        cur = torch.cuda.current_stream(dev)
        for st in hip_streams:
            if st != cur:
                cur.wait_stream(st)       # the chunk is complete when every sub-batch has written its rows
        gatherer.submit(chunk)            # async: overlaps with the next chunk's steps

    def one_step(i, record):
        for j in range(S):
            rp, dp = slot_p[j][i] if (record and coll) else (outs[j][1], outs[j][2])
            _lib.check(L.madrl_pursuit_step(hs[j], act_p[j][i % n_act], None, outs[j][0], rp, dp, outs[j][3], sp[j]))
            if record and coll:
                c, q = divmod(i, CH)
                with torch.cuda.stream(hip_streams[j]):
                    traj[c]["actions"][q][j * per:(j + 1) * per].copy_(actions[i % n_act][j * per:(j + 1) * per])
        if record and coll and (i + 1) % CH == 0:
            submit(traj[i // CH])

    def tail():
        if gatherer is not None:
            if K % CH:
                submit(traj[-1])
            gathered = gatherer.finish()            # episode end: the learner's rank (--gather root) or every rank (all) holds every rank's trajectory
            if gatherer.receives:
                assert sum(t.shape[1] for t in gathered["rewards"]) == K and gathered["rewards"][0].shape[0] == world
            elif gather_mode == "stats":
                # nothing but per-episode statistics crosses xGMI: the episodes this rank finished in the region (returns per pursuer, lengths are
                # the horizon here) -- a few KB, ragged over the ranks
                from madrl_amd.dist import gather_episode_stats
                ended = torch.cat([t["dones"] for t in traj]) != 0
                ret = torch.cat([t["rewards"] for t in traj]).sum(0)[ended.any(0)]
                st = gather_episode_stats(ret, torch.full((ret.shape[0],), H, dtype=torch.int32, device=dev))
                assert len(st["returns"]) == world

    for j, env in enumerate(envs):
        env.reset()
        # steady state: episode ages uniform over [0, H) -- env n reaches the horizon (and runs the fused reset) at step H - age
        env.set_state(dict(t=((torch.arange(per, device=dev, dtype=torch.int32) + j * per) * 7919) % H))
    # ... and the steady state of the kernel's stale-zero masks: which cells outside the map are known to hold 0.0 decides
    # between one 16-byte store and masked 4-byte stores per slot, and that knowledge takes ~2 000 steps to reach its equilibrium -- a
    # launch costs 74 us in the first 500 steps after the masks were reset and 80 us from step 2 000 on (scripts/zmask_drift.py).
    # Untimed, before the W warm-up steps; --prep 0 measures the young state the earlier rounds reported.
    for i in range(args.prep):
        one_step(i, False)
    torch.cuda.synchronize()
    dt, kernel_ms, region_ms = Timer(world, dev, hip_streams, coll=coll).run(one_step, K, W, tail, prepare)
    kernel_kind = envs[0].kernel_kind
    api = None
    if (bool(cpu_budget) if api_leg is None else api_leg) and not coll:
        # The same K-step regions through the DROP-IN API instead of the raw C ABI: BatchedPursuitEvade.step(actions) per sub-batch (as
        # StreamSharded.step(fork=False, join=False), the free-running form the ABI loop above has) -- argument checks, the ctypes call and
        # the result tuple / info dict, whose tensors are views of what the launch wrote (madrl_amd/pursuit.py _step_result).
        from madrl_amd.sharded import StreamSharded
        sh = StreamSharded.from_envs(envs, dev)
        act_parts = [[a[j * per:(j + 1) * per] for j in range(S)] for a in actions]

        def api_step(i, record):
            if S == 1:
                envs[0].step(actions[i % n_act])            # one launch per step, on the current stream
            else:
                sh.step(act_parts[i % n_act], fork=False, join=False)
        adt, akms, aregion = Timer(world, dev, hip_streams).run(api_step, K, min(W, 20))
        api = {"python_api_ms_per_step": adt / K * 1e3, "python_api_region_ms_per_step": [round(x, 6) for x in aregion],
               "python_api_is": "BatchedPursuitEvade.step(actions) on every sub-batch (StreamSharded.step, fork=False, join=False), same actions, same K"}
        del sh
    del envs, outs, hs
    one = None
    if reference_pass and S > 1 and not coll:
        # the same batch as ONE launch per step on one stream, in the same process: what the sub-batch streams are compared with
        one = bench_pursuit(args, variant, K, min(W, 20), rank, world, dev, 0, streams=1, reference_pass=False, api_leg=bool(cpu_budget))
    if rank != 0:
        return None
    bytes_per = algorithmic_bytes_per_env_step(P, E, D, rec_bytes)
    fast = (("pursuit_group_kernel<%%d,%%d,%%d,%%d,%%d,%%d,%d>" % (4 if P * D > 4096 else 2)) if (P + E > 64 or P * D > 2048) else "pursuit_wave_kernel<%d,%d,%d,%d,%d,%d>") % (MS, MS, P, E, R, int(mode["flatten"]))   # (long rows: four wavefronts per env, madrl_amd/build.py pursuit_fast_path)
    kname = fast if kernel_kind == "wave" else "pursuit_kernel<NT>"
    catch = "surround, n_catch 2" if mode["surround"] else "co-location catch, n_catch %d" % mode["n_catch"]
    roof = roofline(bytes_per, N, kernel_ms, dt / K * 1e3, measured_traffic(per, variant, S), kname, streams=S)
    if one is not None:
        roof["one_launch_per_step"] = {k: one["roofline"][k] for k in ("achieved", "frac", "frac_kernel", "frac_vs_measured_copy", "kernel_ms", "algorithmic_bytes_per_launch")}
        roof["one_launch_per_step"]["ms_per_step"] = one["ms_per_step"]
    out = {
        "metric": "env-steps/sec at fixed batch (PursuitEvade %dx%d, %dv%d)" % (MS, MS, P, E),
        "value": world * N * K / dt,
        "unit": "env-steps/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": dt / K * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8/int32 grid state, f32 observations, f64 reward arithmetic",
        "data": "synthetic (uniform random pursuer actions resident in HBM, in-kernel Philox evaders, fused auto-reset; episode ages "
                "start uniform over [0, %d): every step resets ~%d of its %d envs through the two-observation-pass path; %d untimed "
                "steps before the warm-up bring the stale-zero masks of the observation rows to their equilibrium)" % (H, N // H, N, args.prep),
        "config": dict({"workload": "PursuitEvade %dx%d, %d pursuers / %d evaders, obs_range %d, %s, %s, %s, local reward, "
                                    "%d envs per GPU, horizon %d" % (MS, MS, P, E, R, ("synthetic pool of %d maps, sample_maps" % n_pool) if n_pool else "rectangle_map", catch,
                                                                     "flatten" if mode["flatten"] else "(R,R,4) observations", N, H),
                        "envs_per_gpu": N, "envs_total": N * world, "parallelism": "env-sharded x%d" % world,
                        "streams_per_gpu": S, "envs_per_launch": per, "prep_steps": args.prep,
                        "step_is": ("one pass of the step kernel over all %d envs of the GPU: %d launches of %d envs, one per HIP stream, not ordered "
                                    "against each other (independent env instances)" % (N, S, per)) if S > 1 else "one launch of the step kernel over all %d envs" % N,
                        "rccl_ranks": world, "collective_backend": (os.environ.get("MADRL_BENCH_BACKEND", "nccl") if coll else None),
                        "trajectory_gather_in_timed_region": bool(coll), "trajectory_gather": (gather_mode if coll else None),
                        "horizon_resets_per_env_in_timed_region": K / float(H),
                        "horizon_resets_per_step": N / float(H), "horizon_resets_per_launch": per / float(H)}, **region_stats(region_ms)),
        "roofline": roof,
    }
    if api is not None and not cpu_budget:
        out["python_api"] = api
    elif api is not None:
        api["python_api_over_abi"] = api["python_api_ms_per_step"] / out["ms_per_step"]
        if one is not None and "python_api" in one:
            api["python_api_one_launch_ms"] = one["python_api"]["python_api_ms_per_step"]
        out["python_api"] = api
    if cpu_budget:
        c2 = variant == "pursuit"
        attach_cpu_baselines(out, "pursuit" if c2 else None, "pursuit_c1" if c2 else None, lambda: cpu_baseline_port(maps, kw, cpu_budget))
    return out


def bench_rollout(args, K, W, rank, world, dev, cpu_budget=0):
    """Policy in the loop (the sampler loop the reference's runners drive, runners/rurllab.py:298-305; in-tree instance heuristics/pursuit.py:71-85):
    BASELINE configs[1]'s batch, the device chase policy (madrl_amd/heuristics.py) choosing every action from the observation the step
    kernel just wrote, trajectory tensors filled in place, returns scanned at the end of each horizon -- ShardedRolloutCollector over
    sub-batches on their own streams, one captured hipGraph per sub-batch and horizon.  A "step" is one env step of the whole batch
    with everything around it; the horizon is min(K, 50) steps and a timed region is K // horizon collect() calls."""
    import torch
    from madrl_amd.maps import rectangle_map
    from madrl_amd.pursuit import BatchedPursuitEvade
    from madrl_amd.heuristics import PursuitHeuristicPolicy
    from madrl_amd.rollout import ShardedRolloutCollector
    from madrl_amd.sharded import StreamSharded
    MS, P, E, N0, mode = PURSUIT_VARIANTS["pursuit"]
    N, R, H = (args.envs or N0), 7, args.horizon
    S = max(1, int(args.streams)) if args.streams else 1   # step -> policy -> step is one chain: two sub-batches measure the same (102.9 against 102.0 us)
    if N % S or N // S < 64:
        S = 1
    T = max(1, min(K, 50))
    calls = max(1, K // T)
    kw = dict(n_pursuers=P, n_evaders=E, obs_range=R, reward_mech="local", **mode)
    sh = StreamSharded(lambda n_envs, env_id_base, device: BatchedPursuitEvade([rectangle_map(MS, MS)], n_envs=n_envs, device=device, seed=0,
                                                                                env_id_base=rank * N + env_id_base, max_steps=H, auto_reset=True, **kw),
                       N, n_streams=S, device=dev)
    col = ShardedRolloutCollector(sh, [PursuitHeuristicPolicy(R, flatten=True, seed=1, row_id_base=(rank * N + j * (N // S)) * P) for j in range(S)], T,
                                  discount=0.99, graph=True)
    D, rec_bytes = sh.envs[0].obs_dim, sh.envs[0].record_bytes
    for j, env in enumerate(sh.envs):
        env.reset()
        env.set_state(dict(t=((torch.arange(N // S, device=dev, dtype=torch.int32) + j * (N // S)) * 7919) % H))
    for _ in range(max(3, W // T)):   # the first call allocates, the second captures the graphs
        col.collect()

    def step(i, record):
        col.collect()
    dt, kernel_ms, region_ms = Timer(world, dev, None).run(step, calls, 0)
    steps = calls * T
    if rank != 0:
        return None
    # the step kernel's bytes + what the policy needs: the evader channel of every row in, one int32 per pursuer out
    bytes_per = algorithmic_bytes_per_env_step(P, E, D, rec_bytes) + 4 * P * R * R + 4 * P
    ms = dt / steps * 1e3
    region_ms = [x * calls / steps for x in region_ms]
    roof = roofline(bytes_per, N, kernel_ms * calls / steps, ms, measured_traffic(N // S, "pursuit_rollout", S),
                    "pursuit_wave_kernel<16,16,8,30,7,1> + pursuit_policy_kernel (+ gae_kernel per horizon)", streams=S)
    cfg = {"workload": "policy-in-the-loop rollout: PursuitEvade 16x16, 8v30, %d envs per GPU, device chase policy, horizon %d, %d sub-batches, hipGraph per horizon"
                       % (N, T, S), "envs_per_gpu": N, "envs_total": N * world, "parallelism": "env-sharded x%d" % world, "rccl_ranks": world,
           "streams_per_gpu": S, "envs_per_launch": N // S, "horizon": T, "collect_calls_per_region": calls}
    cfg.update(region_stats(region_ms))
    cfg["step_calls_in_process"] = (max(3, W // T) + cfg["timed_regions"] * calls) * T   # what a PMC pass of this command divides its sums by
    out = {"metric": "env-steps/sec with the policy in the loop (PursuitEvade 16x16, 8v30)", "value": world * N * steps / dt, "unit": "env-steps/s", "n_gpus": world,
           "steps": steps, "warmup": W, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "u8/int32 grid state, f32 observations, f64 reward arithmetic", "data": "synthetic (chase policy on the env's own observations)",
           "config": cfg, "roofline": roof}
    if cpu_budget:
        out["cpu_baseline"] = cpu_baseline_rollout(cpu_budget)
    return out


def C_void(v):
    import ctypes
    return ctypes.c_void_p(v)


def bench_other(args, workload, K, W, rank, world, dev, cpu_budget, streams=None, reference_pass=True, api_leg=None):
    """Waterworld (BASELINE configs[2]; `waterworld_std` = the same env under StandardizedEnv, which is how every reference run
    wraps it, runners/run_waterworld.py / run_pursuit.py:57-58), MultiWalker (configs[3]) and the hostage world on the same contract."""
    import numpy as np
    import torch
    from madrl_amd import _lib
    L = _lib.lib()
    extra, flop_per_env_step, live_key, rec_key = {}, None, None, None
    N = args.envs or {"waterworld": 32768, "waterworld_std": 32768, "hostage": 32768, "multiwalker": 16384, "multiwalker_w10": 16384}[workload]
    traffic_key = workload
    S = shard_count(args, N, workload) if streams is None else streams
    per = N // S
    from madrl_amd.sharded import shared_streams
    hip_streams = shared_streams(dev, S) if S > 1 else [torch.cuda.current_stream(dev)]
    sp = [C_void(st.cuda_stream) for st in hip_streams]
    base = lambda j: rank * N + j * per
    if workload in ("waterworld", "waterworld_std"):
        from madrl_amd.waterworld import BatchedMAWaterWorld
        envs = [BatchedMAWaterWorld(5, 10, n_envs=per, device=dev, seed=0, env_id_base=base(j), auto_reset=True, max_blocks=args.max_blocks) for j in range(S)]
        acts = [[(torch.rand((per, 5, 2), device=dev) * 2 - 1).contiguous() for _ in range(8)] for j in range(S)]
        rec = envs[0]._state.numel() // per
        sim_bytes = 40 + 20 + 1 + 8 + 2 * rec                 # actions, rewards, done, info, state record in + out
        n_el = 5 * envs[0].obs_dim
        workload_s = "MAWaterWorld 5 pursuers / 10 evaders / 10 poison / 30 sensors, n_coop 2, %d envs per GPU, timestep_limit 1000" % N
        H = 1000
        if workload == "waterworld":
            outs = [[_lib.ptr(t) for t in (e._obs, e._rew, e._done, e._info)] for e in envs]
            ap = [[_lib.ptr(a) for a in acts[j]] for j in range(S)]
            hs = [e._handle for e in envs]

            def step(i, rec):
                for j in range(S):
                    _lib.check(L.madrl_waterworld_step(hs[j], ap[j][i % 8], None, *outs[j], sp[j]))
            bytes_per = sim_bytes + 4 * n_el
            kernel, binding = "waterworld_kernel<1,5,10,10,30>", "VALU / scalar-pipe issue of the sensing loop (profiles/*_waterworld/pmc_mix.txt), not HBM"
            live_key, rec_key = "waterworld", "waterworld_c3_single_env"
        else:
            from madrl_amd.wrappers import StandardizedEnv
            wenvs = [StandardizedEnv(e, scale_reward=0.5, enable_obsnorm=True, enable_rewnorm=True, fused=args.std_fused) for e in envs]

            def step(i, rec):
                for j in range(S):
                    with torch.cuda.stream(hip_streams[j]):
                        wenvs[j].step(acts[j][i % 8])
            if wenvs[0]._fused:
                # per observation element: float64 mean + var read and written (32 B) + the normalised float32 out (4 B); the raw row never
                # reaches HBM.  Per agent: the reward statistics (32 B) + normalised reward (4 B).  DESIGN.md 4e
                bytes_per = sim_bytes + 36 * n_el + 36 * 5
                kernel = "waterworld_kernel<1,5,10,10,30> with the StandardizedEnv epilogue (madrl_waterworld_set_standardize)"
            else:
                # the raw row is stored by the simulation launch (4 B) and read back by the epilogue launch (4 B) on top of that
                bytes_per = sim_bytes + 44 * n_el + 40 * 5
                kernel = "waterworld_kernel<1,5,10,10,30> + obsnorm_kernel + rewnorm_kernel (wrappers.hip)"
            binding = "HBM: 32 of every 36 - 44 bytes are the per-env float64 running mean / variance (madrl_environments/__init__.py:242-257)"
            workload_s = "StandardizedEnv(obsnorm, rewnorm) around " + workload_s

        def age():  # steady state: episode ages uniform over [0, 1000)
            for j, e in enumerate(envs):
                e.set_state(t=((torch.arange(per, device=dev, dtype=torch.int32) + j * per) * 7919) % H)

        def cpu_fn():
            from oracle import waterworld as ww
            from oracle import pursuit as po
            n = 4096
            o = ww.WaterworldOracle(5, 10, n_envs=n, seed=0, dtype=np.float32)
            o.reset()
            a = np.random.RandomState(0).uniform(-1, 1, (n, 5, 2)).astype(np.float32)
            t0 = time.time(); k = 0
            while time.time() - t0 < cpu_budget:
                o.step(a); k += 1
            dt = time.time() - t0
            return dict(value=n * k / dt, unit="env-steps/s", cores=po.lib().po_num_threads(), kind="port",
                        sample="float32 C oracle (oracle/waterworld_oracle.c, OpenMP; the simulation only, no wrapper), %d envs x %d steps, %.1f s" % (n, k, dt))
    elif workload == "hostage":
        from madrl_amd.hostage import BatchedContinuousHostageWorld
        envs = [BatchedContinuousHostageWorld(3, 10, 5, 2, 2, n_envs=per, device=dev, seed=0, env_id_base=base(j), auto_reset=True,
                                              max_blocks=args.max_blocks) for j in range(S)]
        acts = [[(torch.rand((per, 3, 2), device=dev) * 2 - 1).contiguous() for _ in range(8)] for j in range(S)]
        outs = [[_lib.ptr(t) for t in (e._obs, e._rew, e._done, e._info)] for e in envs]
        ap = [[_lib.ptr(a) for a in acts[j]] for j in range(S)]
        hs = [e._handle for e in envs]

        def step(i, rec):
            for j in range(S):
                _lib.check(L.madrl_hostage_step(hs[j], ap[j][i % 8], None, *outs[j], sp[j]))
        bytes_per = 24 + 4 * 3 * envs[0].obs_dim + 12 + 1 + 8 + 2 * (envs[0]._state.numel() // per)
        kernel, binding = "hostage_kernel<1,3,10,5,30>", "the CU's scalar pipe / VALU issue (profiles/*_hostage/pmc_mix.txt), not HBM"
        workload_s = "ContinuousHostageWorld(3, 10, 5, 2, 2) (hostage.py:483), 30 sensors, %d envs per GPU, timestep_limit 1000" % N
        H = 1000

        def age():
            for j, e in enumerate(envs):
                e.set_state(t=((torch.arange(per, device=dev, dtype=torch.int32) + j * per) * 7919) % H)

        def cpu_fn():
            from oracle import hostage as ho
            from oracle import pursuit as po
            n = 4096
            o = ho.HostageOracle(3, 10, 5, 2, 2, n_envs=n, seed=0, dtype=np.float32)
            o.reset()
            a = np.random.RandomState(0).uniform(-1, 1, (n, 3, 2)).astype(np.float32)
            t0 = time.time(); k = 0
            while time.time() - t0 < cpu_budget:
                _, _, dn, _ = o.step(a); k += 1
                if dn.any():
                    o.reset(mask=dn)
            dt = time.time() - t0
            return dict(value=n * k / dt, unit="env-steps/s", cores=po.lib().po_num_threads(), kind="port",
                        sample="float32 C oracle (oracle/hostage_oracle.c, OpenMP), %d envs x %d steps, %.1f s" % (n, k, dt))
    else:
        from madrl_amd.multiwalker import BatchedMultiWalkerEnv
        H = 500
        # `multiwalker` = BASELINE configs[3] (three walkers); `multiwalker_w10` = the last lesson of the reference's curriculum
        # (lessons/multiwalker/env.yaml: n_walkers 2 .. 10), which runs on the sixteen-lanes-per-env class of the kernels
        MW = 10 if workload == "multiwalker_w10" else 3
        envs = [BatchedMultiWalkerEnv(n_walkers=MW, n_envs=per, device=dev, seed=0, env_id_base=base(j), auto_reset=True,
                                      max_steps=H) for j in range(S)]
        # several sub-batches in flight: the whole b2World::Step of a sub-batch as ONE launch (a wavefront then pays its own collide +
        # solve + continuous-pass time, not the slowest wavefront's of every phase): 3.4 against 4.0 ms per step at four sub-batches;
        # alone on the chip the two forms take the same 4.6 ms (scripts/stream_sweep.sh)
        # (the sixteen-lane class -- 9 / 10 walkers, 4 envs per wavefront -- is the other way round: as three launches its solver runs two
        # wavefronts per SIMD, which the one-launch kernel's registers do not allow: 7.5 against 8.4 ms per step at ten walkers)
        mw_fused = (S > 1 and envs[0].lanes_per_env < 16) if "MADRL_BENCH_MW_FUSED" not in os.environ else os.environ["MADRL_BENCH_MW_FUSED"] == "1"
        if mw_fused:
            for e in envs:
                e.set_mode(fused=True)
        acts = [[(torch.rand((per, MW, 4), device=dev) * 2 - 1).contiguous() for _ in range(8)] for j in range(S)]
        done_rows = torch.zeros((max(K, 1), N), dtype=torch.uint8, device=dev)   # the timed steps write their done bytes here: no extra
        outs = [[_lib.ptr(t) for t in (e._obs, e._rew)] for e in envs]             # launch in the timed region (how many envs ended is counted after it)
        ap = [[_lib.ptr(a) for a in acts[j]] for j in range(S)]
        hs = [e._handle for e in envs]
        dn_p = [[_lib.ptr(done_rows[q][j * per:(j + 1) * per]) for q in range(done_rows.shape[0])] for j in range(S)]
        dn_own = [_lib.ptr(e._done) for e in envs]

        def step(i, rec):
            for j in range(S):
                dn = dn_p[j][i % done_rows.shape[0]] if rec else dn_own[j]
                _lib.check(L.madrl_multiwalker_step(hs[j], ap[j][i % 8], *outs[j], dn, sp[j]))
        # algorithmic HBM bytes per env-step: actions in, observation / reward / done rows out, the per-env record (bodies, joints,
        # contact cache, terrain) read and written once
        bytes_per = 16 * MW + 4 * MW * 32 + 4 * MW + 1 + 2 * envs[0].world_bytes
        kernel = ("mw_step_kernel<all phases> (one launch per sub-batch and step)" if mw_fused else
                  "mw_step_kernel<collide> + <solve> + <continuous pass> (three launches per step; kernel_ms is their sum)")
        # this path is not HBM-bound -- dependent FP32 work of 180 velocity + up to 60 position Gauss-Seidel sweeps
        # over 12 joints and the active manifolds, and the serial sub-steps of the continuous pass, against ~25 KB; a launch ends with
        # its slowest wavefront (16 envs in lockstep), so the binding resource is the latency of the longest per-env chain
        binding = ("latency of the longest per-env chain of dependent FP32 operations (180 + 60 Gauss-Seidel sweeps, time-of-impact sub-steps): "
                   "neither HBM nor VALU throughput (DESIGN.md 4c)")
        if MW == 3:   # the unmodified multi_walker.py over the Box2D shim, timed in the build container (scripts/cpu_reference_bench.py)
            rec_key = "multiwalker_c4_single_env_over_shim"
        flop_per_env_step, flop_src = envs[0].flops_per_env_step()
        extra = {"flop_per_env_step": flop_per_env_step, "flop_source": flop_src}
        workload_s = "MultiWalkerEnv n_walkers=%d, %d envs per GPU, horizon 500 (dynamics: from-scratch Box2D-subset solver, PARITY UNPINNED)" % (MW, N)
        # 200 warm-up steps: episodes last ~60 steps under random actions and all start together, so the first 100 steps see waves of
        # simultaneous falls; after 200 the episode phases of the envs are mixed (the steady state of a rollout)
        K, W = min(K, 50), max(W, 200)

        def age():
            pass   # episodes end by falling long before the horizon; the warm-up above reaches that steady state

        def cpu_fn():
            # the INDEPENDENT plain-C restatement (oracle/multiwalker_ref.c), not the product source compiled for the host
            from oracle import multiwalker_ref as mwr
            n = 1024
            o = mwr.MultiWalkerRef(n_walkers=MW, n_envs=n, seed=0, position_noise=0.0, angle_noise=0.0, poly=True)
            o.reset()
            a = np.random.RandomState(0).uniform(-1, 1, (n, MW, 4)).astype(np.float32)
            t0 = time.time(); k = 0
            while time.time() - t0 < cpu_budget:
                _, _, d = o.step(a); k += 1
                if d.any():
                    o.reset(mask=d)
            dt = time.time() - t0
            return dict(value=n * k / dt, unit="env-steps/s", cores=int(o.L.mwr_num_threads()), kind="port",
                        sample="independent C restatement of MultiWalkerEnv over a Box2D-2.3.0-ordered solver (oracle/multiwalker_ref.c, OpenMP; "
                               "PARITY UNPINNED like the kernel), %d envs x %d steps, %.1f s" % (n, k, dt))
    for j, e in enumerate(envs):
        with torch.cuda.stream(hip_streams[j]):
            (wenvs[j] if workload == "waterworld_std" else e).reset()
    torch.cuda.synchronize()
    age()
    dt, kernel_ms, region_ms = Timer(world, dev, hip_streams).run(step, K, W)
    n_ended = float((done_rows[:K] != 0).sum().item()) / N if workload.startswith("multiwalker") else None
    api = None
    if (bool(cpu_budget) if api_leg is None else api_leg) and workload in ("waterworld", "hostage") and not collective_on(world):
        # the drop-in API instead of the raw C ABI (see bench_pursuit): Batched*.step(action) per sub-batch on its stream
        from madrl_amd.sharded import StreamSharded
        sh = StreamSharded.from_envs(envs, dev)

        def api_step(i, record):
            if S == 1:
                envs[0].step(acts[0][i % 8])
            else:
                sh.step([acts[j][i % 8] for j in range(S)], fork=False, join=False)
        adt, _, aregion = Timer(world, dev, hip_streams).run(api_step, K, min(W, 20))
        api = {"python_api_ms_per_step": adt / K * 1e3, "python_api_region_ms_per_step": [round(x, 6) for x in aregion],
               "python_api_is": "Batched*.step(action) on every sub-batch (StreamSharded.step, fork=False, join=False), same actions, same K"}
        del sh
    del envs
    one = None
    if reference_pass and S > 1 and world == 1:
        one = bench_other(args, workload, K, min(W, 20), rank, world, dev, 0, streams=1, reference_pass=False, api_leg=bool(cpu_budget))
    if rank != 0:
        return None
    roof = roofline(bytes_per, N, kernel_ms, dt / K * 1e3, measured_traffic(per, traffic_key, S), kernel, streams=S, binding_resource=binding)
    if one is not None:
        roof["one_launch_per_step"] = {k: one["roofline"][k] for k in ("achieved", "frac", "frac_kernel", "frac_vs_measured_copy", "kernel_ms", "algorithmic_bytes_per_launch")}
        roof["one_launch_per_step"]["ms_per_step"] = one["ms_per_step"]
    cfg = {"workload": workload_s, "envs_per_gpu": N, "envs_total": N * world, "parallelism": "env-sharded x%d" % world, "rccl_ranks": world,
           "streams_per_gpu": S, "envs_per_launch": per}
    cfg.update(region_stats(region_ms))
    if workload.startswith("multiwalker"):
        tf = flop_per_env_step * N / (kernel_ms * 1e-3) / 1e12
        roof.update({"valu_flops_achieved_TFLOPs": tf, "valu_peak_TFLOPs": VALU_PEAK_TFLOPS, "valu_frac": tf / VALU_PEAK_TFLOPS, **extra})
        cfg["episode_ends_per_env_in_last_region"] = n_ended
    else:
        cfg["horizon_resets_per_env_in_timed_region"] = K / float(H)
    out = {"metric": "env-steps/sec at fixed batch (%s)" % workload, "value": world * N * K / dt, "unit": "env-steps/s",
           "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32" if workload != "waterworld_std" else "f32 simulation, f64 running statistics",
           "data": "synthetic (uniform random actions resident in HBM, in-kernel Philox, fused auto-reset, steady-state episode ages)",
           "config": cfg, "roofline": roof}
    if api is not None:
        if cpu_budget:
            api["python_api_over_abi"] = api["python_api_ms_per_step"] / out["ms_per_step"]
            if one is not None and "python_api" in one:
                api["python_api_one_launch_ms"] = one["python_api"]["python_api_ms_per_step"]
        out["python_api"] = api
    if cpu_budget:
        attach_cpu_baselines(out, live_key, rec_key, cpu_fn)
    return out


def _r(x, nd=4):
    """numbers of the printed line: 6 significant digits are plenty and keep it short"""
    return float("%.6g" % x) if isinstance(x, float) else x


def compact_roofline(r, side=False):
    keep = ["bound", "achieved", "peak", "unit", "frac", "frac_kernel", "traffic", "kernel", "kernel_ms"]
    out = {k: _r(r[k]) for k in keep if k in r}
    if "one_launch_per_step" in r:
        out["one_launch_ms"] = _r(r["one_launch_per_step"]["ms_per_step"])
        out["one_launch_frac"] = _r(r["one_launch_per_step"]["frac"])
    for k in ("valu_frac", "flop_per_env_step"):
        if k in r:
            out[k] = _r(r[k])
    if side:   # (achieved = frac x the headline's peak)
        out.pop("peak", None); out.pop("unit", None); out.pop("bound", None); out.pop("achieved", None)
    return out


def compact_line(out):
    """The ONE JSON line rank 0 prints: the contract's fields, short enough (a few KB) that a log tail holds every workload.  What the
    fields mean is in DESIGN.md "Measurement"; the long form of this record (every key of the rounds before, prose included) goes to
    --full-record / gpurun_out/bench_full.json."""
    line = {k: _r(out[k]) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype")}
    line["data"] = "synthetic"
    c = out["config"]
    line["config"] = {k: c[k] for k in ("workload", "envs_per_gpu", "envs_total", "parallelism", "streams_per_gpu", "rccl_ranks", "collective_backend",
                                        "trajectory_gather_in_timed_region", "trajectory_gather") if k in c}
    line["config"]["region_ms_per_step"] = [_r(float(x)) for x in c["region_ms_per_step"]]
    line["roofline"] = compact_roofline(out["roofline"])
    if "python_api" in out:   # the same regions through Batched*.step() instead of the raw C ABI
        for k in ("python_api_ms_per_step", "python_api_one_launch_ms"):
            if k in out["python_api"]:
                line[k] = _r(float(out["python_api"][k]))
    if "cpu_baseline" in out:
        b = out["cpu_baseline"]
        line["cpu_baseline"] = {"value": _r(float(b["value"])), "unit": b["unit"], "cores": b["cores"], "kind": b["kind"], "sample": b["sample"][:160]}
    if "cpu_reference_recorded" in out:
        b = out["cpu_reference_recorded"]
        line["cpu_reference_recorded"] = {"value": _r(float(b["value"])), "cores": b["cores"], "kind": b["kind"], "host": b["host"][:80]}
    if "workloads" in out:
        line["workloads"] = {}
        for name, w in out["workloads"].items():
            if "error" in w:
                line["workloads"][name] = {"error": w["error"][:200]}
                continue
            e = {"value": _r(float(w["value"])), "ms_per_step": _r(float(w["ms_per_step"])), "steps": w["steps"], "workload": w["config"]["workload"][:56],
                 "envs": w["config"]["envs_per_gpu"], "streams": w["config"]["streams_per_gpu"], "roofline": compact_roofline(w["roofline"], side=True)}
            if "cpu_baseline" in w:
                e["cpu_baseline"] = {"value": _r(float(w["cpu_baseline"]["value"])), "cores": w["cpu_baseline"]["cores"], "kind": w["cpu_baseline"]["kind"]}
            if "python_api" in w:
                e["python_api_ms"] = _r(float(w["python_api"]["python_api_ms_per_step"]))
            if "over_headline" in w:   # pursuit_collective: the N > 1 code path (trajectory recording + RCCL exchange) in a one-rank group
                e["over_headline"] = _r(float(w["over_headline"]))
                e["gather"] = w["config"].get("trajectory_gather")
            line["workloads"][name] = e
    return line


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU (the driver's own
    command line does the same thing explicitly)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


WORKLOADS = ["pursuit", "pursuit_c5", "pursuit_colocate", "pursuit_authors", "pursuit_rollout", "waterworld", "waterworld_std", "multiwalker", "multiwalker_w10", "hostage"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--envs", type=int, default=0, help="env instances per GPU (0 = the BASELINE config's batch)")
    ap.add_argument("--workload", default="pursuit", choices=WORKLOADS,
                    help="pursuit = BASELINE.json's metric (default, configs[1]); pursuit_c5 = configs[4]'s per-GPU shard "
                         "(32x32, 16 v 60, 32 768 envs: `--workload pursuit_c5 --gpus 8` IS configs[4], 262 144 envs); pursuit_colocate = the "
                         "survey's secondary catch mode; waterworld_std = Waterworld under StandardizedEnv; the others are the remaining north_star envs")
    ap.add_argument("--streams", type=int, default=0, help="sub-batches per GPU, each stepped on its own HIP stream (madrl_amd/sharded.py); 1 = one launch "
                                                           "per step; 0 (default) = the measured best per workload (DEFAULT_STREAMS)")
    ap.add_argument("--std-fused", type=lambda v: None if v == "auto" else v not in ("0", "false", "no"), default=False,
                    help="waterworld_std: StandardizedEnv fused into the step kernel (default: the stand-alone epilogue kernels, which overlap with "
                         "the other sub-batch's simulation launch; auto = the wrapper's own default, fused)")
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--max-blocks", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather", default="root", choices=("root", "all", "stats"),
                    help="N > 1: who receives the compact trajectory (actions, rewards, dones) of the timed region, exchanged in chunks that overlap with the "
                         "stepping -- root = rank 0 only (default: north_star's gather at episode end, the sampler workers returning their paths to one learner), "
                         "all = every rank (all-gather: a learner data-parallel over the same ranks), stats = nobody (per-episode returns only)")
    ap.add_argument("--full", action="store_true", help="print the long record (every key, prose included) instead of the compact line")
    ap.add_argument("--full-record", default="", help="also write the long record to this file (default: gpurun_out/bench_full.json when that directory exists)")
    ap.add_argument("--no-workloads", action="store_true", help="headline only: do not time the other BASELINE configs")
    ap.add_argument("--horizon", type=int, default=500, help="max_path_length (runners/__init__.py:88)")
    ap.add_argument("--prep", type=int, default=2000, help="Pursuit: untimed steps before the warm-up that bring the stale-zero masks of the observation "
                                                            "rows to their equilibrium (a rollout's steady state, like the spread episode ages)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)   # does not return
    # The contract: rank 0 prints ONE JSON line.  Libraries write to file descriptor 1 behind Python's back -- RCCL prints a version banner
    # through C stdio when its first communicator comes up, and a redirected C stream is flushed at process exit, i.e. AFTER the line.  So
    # descriptor 1 is pointed at stderr for the life of the process and the line goes to a private copy of the real stdout.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    backend = os.environ.get("MADRL_BENCH_BACKEND", "nccl")  # "gloo": exercise the N > 1 path with all ranks on one GPU (tests only)
    if backend == "gloo":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if collective_on(world):
        ensure_group(dev)   # the launcher's group, or (MADRL_BENCH_FORCE_COLLECTIVE with one rank) a one-rank group with its own rendezvous

    K, W = args.steps, args.warmup
    cpu = (not args.no_cpu_baseline) and not collective_on(world)   # the CPU baselines are reported with the plain 1-GPU line only
    scale = float(os.environ.get("MADRL_BENCH_CPU_BUDGET", "1"))   # tests shorten the CPU samples
    head_cpu, side_cpu = (10.0 * scale, 3.0 * scale) if cpu else (0, 0)
    if args.workload == "pursuit_rollout":
        out = bench_rollout(args, K, W, rank, world, dev, head_cpu)
    elif args.workload.startswith("pursuit"):
        out = bench_pursuit(args, args.workload, K, W, rank, world, dev, head_cpu)
    else:
        out = bench_other(args, args.workload, K, W, rank, world, dev, head_cpu)
    # The same driver run times every other BASELINE config (N = 1, default workload, default batch): bounded steps and a bounded
    # CPU sample (3 s) each
    if args.workload == "pursuit" and not collective_on(world) and not args.no_workloads and not args.envs:
        wl = {}
        # (their own step counts, whatever --steps says: the contract's K is the headline's; a 20-step region of a two-stream pipeline is a third
        # fill and drain -- Waterworld reads 51 us per step at K = 20 and 42 at K = 200)
        for name, k, w in (("waterworld", 200, 20), ("multiwalker", 50, 20), ("pursuit_c5", 200, 20), ("pursuit_colocate", 200, 20), ("waterworld_std", 100, 20),
                           ("multiwalker_w10", 20, 20), ("pursuit_rollout", 200, 20), ("hostage", 200, 20), ("pursuit_authors", 100, 20)):
            try:
                r = bench_rollout(args, k, w, rank, world, dev, side_cpu) if name == "pursuit_rollout" else \
                    bench_pursuit(args, name, k, w, rank, world, dev, side_cpu) if name.startswith("pursuit") else \
                    bench_other(args, name, k, w, rank, world, dev, side_cpu)
                wl[name] = {f: r[f] for f in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline", "cpu_baseline", "python_api") if f in r}
            except Exception as e:  # a failing side workload must not take the headline down; it shows up as an error entry
                wl[name] = {"error": repr(e)}
        # What the N > 1 code path costs, measured where it can be: the headline config again with the trajectory recording and the RCCL
        # exchange (--gather) in the timed region, in a ONE-rank group (two ranks on one device are refused by RCCL).  The collectives are then
        # device-local copies made by the backend: this prices the recording, the chunking and the backend's launches, not xGMI.
        try:
            import torch.distributed as dist
            made = ensure_group(dev)
            r = bench_pursuit(args, "pursuit", 200, 20, rank, world, dev, 0, reference_pass=False, api_leg=False, coll=True)
            wl["pursuit_collective"] = {f: r[f] for f in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline") if f in r}
            wl["pursuit_collective"]["over_headline"] = r["ms_per_step"] / out["ms_per_step"]
            if made:
                dist.destroy_process_group()
        except Exception as e:
            wl["pursuit_collective"] = {"error": repr(e)}
        out["workloads"] = wl
    if rank == 0:
        full_path = args.full_record or (os.path.join(ROOT, "gpurun_out", "bench_full.json") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else None)
        if full_path:
            try:
                with open(full_path, "w") as f:
                    json.dump(out, f, indent=1)
            except OSError:
                pass
        json_out.write(json.dumps(out if args.full else compact_line(out), separators=(",", ":")) + "\n")
        json_out.flush()
    if collective_on(world):
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
