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
all-gathered over RCCL at the end, inside the timed region.

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

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md)
HBM_COPY_GBPS = 6290.0  # the same guide's measured copy rate (MI355X_MICROARCH.md:34-35): what a pure streaming kernel reaches
VALU_PEAK_TFLOPS = 157.3  # FP32 vector peak, same guide
REPEATS = 3             # timed K-step regions per workload (each bracketed by barrier + synchronize); the line reports the median


def algorithmic_bytes_per_env_step(P, E, D, rec_bytes):
    """DESIGN.md "Algorithmic bytes": actions in, obs/rewards/done/removed out, packed state
    record read + written once."""
    return 4 * P + 4 * P * D + 4 * P + 1 + 4 + 2 * rec_bytes


def measured_traffic(n_envs, workload="pursuit"):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/*/pmc_traffic.json,
    FETCH_SIZE + WRITE_SIZE collected in separate runs by scripts/profile.sh); None if absent or
    taken at another batch size.  bench.py cannot collect PMC counters on itself."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_traffic.json"))):
        try:
            j = json.load(open(f))
        except Exception:
            continue
        if int(j.get("envs", -1)) == int(n_envs) and j.get("workload", "pursuit") == workload:
            best = (float(j["traffic_bytes_per_launch"]), os.path.relpath(f, ROOT))
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


class Timer(object):
    """the bench contract: W untimed warm-up steps, then EXACTLY K steps bracketed by a barrier + synchronize on both sides;
    HIP events on the launch stream give the average launch duration"""

    def __init__(self, world, dev):
        self.world, self.dev = world, dev

    def barrier(self):
        import torch
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def region(self, step, K, tail=None):
        import torch
        self.barrier()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for i in range(K):
            step(i, True)
        ev1.record()  # events bracket the K step launches (and the small trajectory copies when N > 1)
        if tail is not None:
            tail()
        self.barrier()
        dt = time.perf_counter() - t0
        kernel_ms = ev0.elapsed_time(ev1) / K  # average launch duration incl. inter-launch gaps
        if self.world > 1:
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


def roofline(bytes_per, N, kernel_ms, traffic, kernel, **more):
    achieved = bytes_per * N / (kernel_ms * 1e-3) / 1e9
    out = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
           "frac_vs_measured_copy": achieved / HBM_COPY_GBPS, "measured_copy_peak": HBM_COPY_GBPS,
           "traffic": traffic[0] if traffic else None, "traffic_source": traffic[1] if traffic else None,
           "algorithmic_bytes_per_launch": bytes_per * N, "kernel": kernel, "kernel_ms": kernel_ms,
           "algorithmic_bytes_per_env_step": bytes_per}
    out.update(more)
    return out


PURSUIT_VARIANTS = {
    # name: (map side, pursuers, evaders, default envs per GPU, env kwargs)
    "pursuit": (16, 8, 30, 65536, dict(n_catch=2, surround=True, flatten=True)),              # BASELINE configs[1]
    "pursuit_c5": (32, 16, 60, 32768, dict(n_catch=2, surround=True, flatten=True)),          # configs[4], one GPU's shard
    # SURVEY 8 preamble's secondary mode: heuristics/pursuit.py:66-67 (co-location catch, (R, R, 4) observations)
    "pursuit_colocate": (16, 8, 30, 65536, dict(n_catch=4, surround=False, flatten=False)),
}


def bench_pursuit(args, variant, K, W, rank, world, dev, cpu_budget):
    import numpy as np
    import torch
    from madrl_amd.maps import rectangle_map
    from madrl_amd.pursuit import BatchedPursuitEvade
    from madrl_amd import _lib
    MS, P, E, N0, mode = PURSUIT_VARIANTS[variant]
    N, R = (args.envs or N0), 7
    H = args.horizon
    maps = [rectangle_map(MS, MS)]
    kw = dict(n_pursuers=P, n_evaders=E, obs_range=R, reward_mech="local", **mode)
    env = BatchedPursuitEvade(maps, n_envs=N, device=dev, seed=0, env_id_base=rank * N, max_steps=H,
                              auto_reset=True, threads=args.threads, max_blocks=args.max_blocks, **kw)
    D = env.obs_dim
    rec_bytes = env.record_bytes
    gen = torch.Generator(device=dev).manual_seed(rank)
    n_act = 16
    actions = [torch.randint(0, 5, (N, P), generator=gen, device=dev, dtype=torch.int32) for _ in range(n_act)]
    # compact trajectory of the timed region (what a sampler returns to the learner), cut into
    # chunks whose all-gather over RCCL/xGMI overlaps with the stepping of the next chunk
    CH = 50
    n_chunks = (K + CH - 1) // CH
    chunk_len = [min(CH, K - c * CH) for c in range(n_chunks)]
    traj = [dict(actions=torch.zeros((chunk_len[c], N, P), dtype=torch.uint8, device=dev),
                 rewards=torch.zeros((chunk_len[c], N, P), dtype=torch.float32, device=dev),
                 dones=torch.zeros((chunk_len[c], N), dtype=torch.uint8, device=dev)) for c in range(n_chunks)] if world > 1 else []
    L = _lib.lib()
    h = env._handle
    obs_p, rew_p, done_p, rem_p = (_lib.ptr(t) for t in (env._obs, env._rew, env._done, env._removed))
    act_p = [_lib.ptr(a) for a in actions]
    # N > 1: the step kernel writes rewards / dones straight into their slot of the trajectory chunk (the C ABI takes any
    # device pointer), only the action row is copied (int32 -> uint8)
    slot_p = [(_lib.ptr(traj[c]["rewards"][j]), _lib.ptr(traj[c]["dones"][j])) for c in range(n_chunks) for j in range(chunk_len[c])] if world > 1 else []
    gatherer = None
    if world > 1:
        from madrl_amd.dist import ChunkedTrajectoryGather
        gatherer = ChunkedTrajectoryGather()

    def prepare():
        if gatherer is not None:
            gatherer.reserve(traj)   # receive buffers + RCCL channel setup stay out of the timed region

    def one_step(i, record):
        rp, dp = slot_p[i] if (record and world > 1) else (rew_p, done_p)
        _lib.check(L.madrl_pursuit_step(h, act_p[i % n_act], None, obs_p, rp, dp, rem_p, _lib.current_stream(dev)))
        if record and world > 1:
            c, j = divmod(i, CH)
            traj[c]["actions"][j].copy_(actions[i % n_act])
            if (i + 1) % CH == 0:
                gatherer.submit(traj[i // CH])      # async: overlaps with the next chunk's steps

    def tail():
        if gatherer is not None:
            if K % CH:
                gatherer.submit(traj[-1])
            gathered = gatherer.finish()            # episode end: every rank holds every rank's trajectory
            assert sum(t.shape[1] for t in gathered["rewards"]) == K and gathered["rewards"][0].shape[0] == world

    env.reset()
    # steady state: episode ages uniform over [0, H) -- env n reaches the horizon (and runs the fused reset) at step H - age
    env.set_state(dict(t=(torch.arange(N, device=dev, dtype=torch.int32) * 7919) % H))
    dt, kernel_ms, region_ms = Timer(world, dev).run(one_step, K, W, tail, prepare)
    if rank != 0:
        return None
    bytes_per = algorithmic_bytes_per_env_step(P, E, D, rec_bytes)
    fast = ("pursuit_group_kernel<%d,%d,%d,%d,%d,%d,2>" if P + E > 64 else "pursuit_wave_kernel<%d,%d,%d,%d,%d,%d>") % (MS, MS, P, E, R, int(mode["flatten"]))
    kname = fast if env.kernel_kind == "wave" else "pursuit_kernel<NT>"
    catch = "surround, n_catch 2" if mode["surround"] else "co-location catch, n_catch %d" % mode["n_catch"]
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
                "start uniform over [0, %d): every launch resets ~%d of its %d envs through the two-observation-pass path)" % (H, N // H, N),
        "config": dict({"workload": "PursuitEvade %dx%d rectangle_map, %d pursuers / %d evaders, obs_range 7, %s, %s, local reward, "
                                    "%d envs per GPU, horizon %d" % (MS, MS, P, E, catch, "flatten" if mode["flatten"] else "(R,R,4) observations", N, H),
                        "envs_per_gpu": N, "envs_total": N * world, "parallelism": "env-sharded x%d" % world,
                        "rccl_ranks": world, "collective_backend": (os.environ.get("MADRL_BENCH_BACKEND", "nccl") if world > 1 else None),
                        "horizon_resets_per_env_in_timed_region": K / float(H),
                        "horizon_resets_per_launch": N / float(H)}, **region_stats(region_ms)),
        "roofline": roofline(bytes_per, N, kernel_ms, measured_traffic(N, variant), kname),
    }
    if cpu_budget:
        c2 = variant == "pursuit"
        attach_cpu_baselines(out, "pursuit" if c2 else None, "pursuit_c1" if c2 else None, lambda: cpu_baseline_port(maps, kw, cpu_budget))
    del env
    return out


def bench_other(args, workload, K, W, rank, world, dev, cpu_budget):
    """Waterworld (BASELINE configs[2]; `waterworld_std` = the same env under StandardizedEnv, which is how every reference run
    wraps it, runners/run_waterworld.py / run_pursuit.py:57-58), MultiWalker (configs[3]) and the hostage world on the same contract."""
    import numpy as np
    import torch
    from madrl_amd import _lib
    L = _lib.lib()
    extra, flop_per_env_step, live_key, rec_key = {}, None, None, None
    if workload in ("waterworld", "waterworld_std"):
        from madrl_amd.waterworld import BatchedMAWaterWorld
        N = args.envs or 32768
        env = BatchedMAWaterWorld(5, 10, n_envs=N, device=dev, seed=0, env_id_base=rank * N, auto_reset=True,
                                  max_blocks=args.max_blocks)
        acts = [(torch.rand((N, 5, 2), device=dev) * 2 - 1).contiguous() for _ in range(8)]
        rec = env._state.numel() // N
        sim_bytes = 40 + 20 + 1 + 8 + 2 * rec                 # actions, rewards, done, info, state record in + out
        n_el = 5 * env.obs_dim
        workload_s = "MAWaterWorld 5 pursuers / 10 evaders / 10 poison / 30 sensors, n_coop 2, %d envs per GPU, timestep_limit 1000" % N
        H = 1000
        if workload == "waterworld":
            outs = [_lib.ptr(t) for t in (env._obs, env._rew, env._done, env._info)]
            step = lambda i, rec: _lib.check(L.madrl_waterworld_step(env._handle, _lib.ptr(acts[i % 8]), None, *outs, _lib.current_stream(dev)))
            bytes_per = sim_bytes + 4 * n_el
            kernel, binding = "waterworld_kernel<1,5,10,10,30>", "VALU / scalar-pipe issue of the sensing loop (profiles/*_waterworld/pmc_mix.txt), not HBM"
            live_key, rec_key = "waterworld", "waterworld_c3_single_env"
        else:
            from madrl_amd.wrappers import StandardizedEnv
            wenv = StandardizedEnv(env, scale_reward=0.5, enable_obsnorm=True, enable_rewnorm=True)
            assert wenv._fused, "the Waterworld engine fuses StandardizedEnv into its step kernel"
            step = lambda i, rec: wenv.step(acts[i % 8])
            # per observation element: float64 mean + var read and written (32 B) + the normalised float32 out (4 B); the raw row never
            # reaches HBM.  Per agent: the reward statistics (32 B) + normalised reward (4 B).  DESIGN.md 4e
            bytes_per = sim_bytes + 36 * n_el + 36 * 5
            kernel = "waterworld_kernel<1,5,10,10,30> with the StandardizedEnv epilogue (madrl_waterworld_set_standardize)"
            binding = "HBM: 32 of every 36 bytes are the per-env float64 running mean / variance (madrl_environments/__init__.py:242-257)"
            workload_s = "StandardizedEnv(obsnorm, rewnorm) around " + workload_s

        def age():  # steady state: episode ages uniform over [0, 1000)
            env.set_state(t=(torch.arange(N, device=dev, dtype=torch.int32) * 7919) % H)

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
        N = args.envs or 32768
        env = BatchedContinuousHostageWorld(3, 10, 5, 2, 2, n_envs=N, device=dev, seed=0, env_id_base=rank * N, auto_reset=True,
                                            max_blocks=args.max_blocks)
        acts = [(torch.rand((N, 3, 2), device=dev) * 2 - 1).contiguous() for _ in range(8)]
        outs = [_lib.ptr(t) for t in (env._obs, env._rew, env._done, env._info)]
        step = lambda i, rec: _lib.check(L.madrl_hostage_step(env._handle, _lib.ptr(acts[i % 8]), None, *outs, _lib.current_stream(dev)))
        bytes_per = 24 + 4 * 3 * env.obs_dim + 12 + 1 + 8 + 2 * (env._state.numel() // N)
        kernel, binding = "hostage_kernel<1,3,10,5,30>", "the CU's scalar pipe / VALU issue (profiles/*_hostage/pmc_mix.txt), not HBM"
        workload_s = "ContinuousHostageWorld(3, 10, 5, 2, 2) (hostage.py:483), 30 sensors, %d envs per GPU, timestep_limit 1000" % N
        H = 1000

        def age():
            env.set_state(t=(torch.arange(N, device=dev, dtype=torch.int32) * 7919) % H)

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
        N = args.envs or 16384
        H = 500
        env = BatchedMultiWalkerEnv(n_walkers=3, n_envs=N, device=dev, seed=0, env_id_base=rank * N, auto_reset=True,
                                    max_steps=H, max_blocks=args.max_blocks)
        acts = [(torch.rand((N, 3, 4), device=dev) * 2 - 1).contiguous() for _ in range(8)]
        done_rows = torch.zeros((max(K, 1), N), dtype=torch.uint8, device=dev)   # the timed steps write their done bytes here: no extra
        outs = [_lib.ptr(t) for t in (env._obs, env._rew)]                       # launch in the timed region (how many envs ended is counted after it)

        def step(i, rec):
            dn = done_rows[i % done_rows.shape[0]] if rec else env._done
            _lib.check(L.madrl_multiwalker_step(env._handle, _lib.ptr(acts[i % 8]), *outs, _lib.ptr(dn), _lib.current_stream(dev)))
        # algorithmic HBM bytes per env-step: actions in, observation / reward / done rows out, the per-env record (bodies, joints,
        # contact cache, terrain) read and written once
        bytes_per = 48 + 4 * 3 * 32 + 12 + 1 + 2 * env.world_bytes
        kernel = "mw_step_kernel<collide> + <solve> + <continuous pass> (three launches per step; kernel_ms is their sum)"
        # this path is not HBM-bound -- dependent FP32 work of 180 velocity + up to 60 position Gauss-Seidel sweeps
        # over 12 joints and the active manifolds, and the serial sub-steps of the continuous pass, against ~25 KB; a launch ends with
        # its slowest wavefront (16 envs in lockstep), so the binding resource is the latency of the longest per-env chain
        binding = ("latency of the longest per-env chain of dependent FP32 operations (180 + 60 Gauss-Seidel sweeps, time-of-impact sub-steps): "
                   "neither HBM nor VALU throughput (DESIGN.md 4c)")
        flop_per_env_step, flop_src = env.flops_per_env_step()
        extra = {"flop_per_env_step": flop_per_env_step, "flop_source": flop_src}
        workload_s = "MultiWalkerEnv n_walkers=3, %d envs per GPU, horizon 500 (dynamics: from-scratch Box2D-subset solver, PARITY UNPINNED)" % N
        # 200 warm-up steps: episodes last ~60 steps under random actions and all start together, so the first 100 steps see waves of
        # simultaneous falls; after 200 the episode phases of the envs are mixed (the steady state of a rollout)
        K, W = min(K, 50), max(W, 200)

        def age():
            pass   # episodes end by falling long before the horizon; the warm-up above reaches that steady state

        def cpu_fn():
            # the INDEPENDENT plain-C restatement (oracle/multiwalker_ref.c), not the product source compiled for the host
            from oracle import multiwalker_ref as mwr
            n = 1024
            o = mwr.MultiWalkerRef(n_walkers=3, n_envs=n, seed=0, position_noise=0.0, angle_noise=0.0, poly=True)
            o.reset()
            a = np.random.RandomState(0).uniform(-1, 1, (n, 3, 4)).astype(np.float32)
            t0 = time.time(); k = 0
            while time.time() - t0 < cpu_budget:
                _, _, d = o.step(a); k += 1
                if d.any():
                    o.reset(mask=d)
            dt = time.time() - t0
            return dict(value=n * k / dt, unit="env-steps/s", cores=int(o.L.mwr_num_threads()), kind="port",
                        sample="independent C restatement of MultiWalkerEnv over a Box2D-2.3.0-ordered solver (oracle/multiwalker_ref.c, OpenMP; "
                               "PARITY UNPINNED like the kernel), %d envs x %d steps, %.1f s" % (n, k, dt))
    env.reset()
    age()
    dt, kernel_ms, region_ms = Timer(world, dev).run(step, K, W)
    if rank != 0:
        return None
    roof = roofline(bytes_per, N, kernel_ms, measured_traffic(N, workload), kernel, binding_resource=binding)
    cfg = {"workload": workload_s, "envs_per_gpu": N, "envs_total": N * world, "parallelism": "env-sharded x%d" % world, "rccl_ranks": world}
    cfg.update(region_stats(region_ms))
    if workload == "multiwalker":
        tf = flop_per_env_step * N / (kernel_ms * 1e-3) / 1e12
        roof.update({"valu_flops_achieved_TFLOPs": tf, "valu_peak_TFLOPs": VALU_PEAK_TFLOPS, "valu_frac": tf / VALU_PEAK_TFLOPS, **extra})
        cfg["episode_ends_per_env_in_last_region"] = float((done_rows[:K] != 0).sum().item()) / N
    else:
        cfg["horizon_resets_per_env_in_timed_region"] = K / float(H)
    out = {"metric": "env-steps/sec at fixed batch (%s)" % workload, "value": world * N * K / dt, "unit": "env-steps/s",
           "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32" if workload != "waterworld_std" else "f32 simulation, f64 running statistics",
           "data": "synthetic (uniform random actions resident in HBM, in-kernel Philox, fused auto-reset, steady-state episode ages)",
           "config": cfg, "roofline": roof}
    if cpu_budget:
        attach_cpu_baselines(out, live_key, rec_key, cpu_fn)
    del env
    return out


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


WORKLOADS = ["pursuit", "pursuit_c5", "pursuit_colocate", "waterworld", "waterworld_std", "multiwalker", "hostage"]


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
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--max-blocks", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-workloads", action="store_true", help="headline only: do not time the other BASELINE configs")
    ap.add_argument("--horizon", type=int, default=500, help="max_path_length (runners/__init__.py:88)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)   # does not return
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
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    K, W = args.steps, args.warmup
    cpu = (not args.no_cpu_baseline) and world == 1   # the CPU baselines are reported with the 1-GPU line only
    scale = float(os.environ.get("MADRL_BENCH_CPU_BUDGET", "1"))   # tests shorten the CPU samples
    head_cpu, side_cpu = (10.0 * scale, 3.0 * scale) if cpu else (0, 0)
    if args.workload.startswith("pursuit"):
        out = bench_pursuit(args, args.workload, K, W, rank, world, dev, head_cpu)
    else:
        out = bench_other(args, args.workload, K, W, rank, world, dev, head_cpu)
    # The same driver run times every other BASELINE config (N = 1, default workload, default batch): bounded steps and a bounded
    # CPU sample (3 s) each
    if args.workload == "pursuit" and world == 1 and not args.no_workloads and not args.envs:
        wl = {}
        for name, k, w in (("waterworld", min(K, 200), min(W, 20)), ("multiwalker", min(K, 50), W), ("pursuit_c5", min(K, 200), min(W, 20)),
                           ("pursuit_colocate", min(K, 200), min(W, 20)), ("waterworld_std", min(K, 100), min(W, 20))):
            try:
                r = bench_pursuit(args, name, k, w, rank, world, dev, side_cpu) if name.startswith("pursuit") else \
                    bench_other(args, name, k, w, rank, world, dev, side_cpu)
                wl[name] = {f: r[f] for f in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline", "cpu_baseline") if f in r}
            except Exception as e:  # a failing side workload must not take the headline down; it shows up as an error entry
                wl[name] = {"error": repr(e)}
        out["workloads"] = wl
    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
