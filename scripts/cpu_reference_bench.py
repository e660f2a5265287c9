#!/usr/bin/env python
"""Time the UNMODIFIED reference's NumPy CPU path (BASELINE.json configs[0], SURVEY.md §8(d) "C1 input"
and "CPU reference timing") and write profiles/<round>_cpu_reference/record.json.

    python scripts/cpu_reference_bench.py [--steps 1500] [--out profiles/r02_cpu_reference/record.json]

The reference is imported from /root/reference through oracle/ref_loader.py (gym / rltools shims only, no source
change); it does not exist on the GPU box, so this script runs in the build container and bench.py only QUOTES the
committed record (labelled with the host it was taken on) next to the live timing of the C port.

Workloads, driven the way heuristics/pursuit.py:64-85 drives the env (reset, then step until done or max_path_length
= 500, runners/__init__.py:88):
  pursuit     PursuitEvade([rectangle_map(16,16)], n_evaders=30, n_pursuers=8, obs_range=7, n_catch=2, surround=True,
              flatten=True, reward_mech='local'); np.random.seed(0); actions RandomState(0).randint(5, size=8)
              (pursuit_evade.py:209-262)
  waterworld  MAWaterWorld(5, 10) defaults; actions U(-1, 1) [5, 2] (waterworld.py:220-436)
  multiwalker MultiWalkerEnv(n_walkers=3) over the Box2D shim (round 5; see run_multiwalker)
each as 1 process and as multiprocessing.Pool(os.cpu_count()) with one env per process, OMP_NUM_THREADS=1.
"""
import argparse
import json
import os
import platform
import sys
import time

os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
os.environ.setdefault("MKL_NUM_THREADS", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def run_pursuit(arg):
    seed, steps = arg
    import numpy as np
    from oracle import ref_loader
    ref = ref_loader.load()
    np.random.seed(seed)
    env = ref["PursuitEvade"]([ref["TwoDMaps"].rectangle_map(16, 16)], n_evaders=30, n_pursuers=8, obs_range=7,
                              n_catch=2, surround=True, flatten=True, reward_mech="local")
    rng = np.random.RandomState(seed)
    env.reset()
    for _ in range(20):  # warm-up (imports, first-touch)
        env.step(rng.randint(5, size=8))
    env.reset()
    t_in = 0
    n_resets = 0
    t0 = time.perf_counter()
    for _ in range(steps):
        _, _, done, _ = env.step(rng.randint(5, size=8))
        t_in += 1
        if done or t_in >= 500:
            env.reset()
            t_in = 0
            n_resets += 1
    return steps, time.perf_counter() - t0, n_resets


def run_waterworld(arg):
    seed, steps = arg
    import numpy as np
    from oracle import ref_loader
    ref = ref_loader.load()
    np.random.seed(seed)
    env = ref["MAWaterWorld"](5, 10)
    env.seed(seed)
    rng = np.random.RandomState(seed)
    env.reset()
    for _ in range(20):
        env.step(rng.uniform(-1, 1, (5, 2)))
    env.reset()
    t_in = 0
    n_resets = 0
    t0 = time.perf_counter()
    for _ in range(steps):
        _, _, done, _ = env.step(rng.uniform(-1, 1, (5, 2)))
        t_in += 1
        if done or t_in >= 1000:
            env.reset()
            t_in = 0
            n_resets += 1
    return steps, time.perf_counter() - t0, n_resets


def run_multiwalker(arg):
    """the UNMODIFIED multi_walker.py (n_walkers = 3, BASELINE configs[3]) -- over the test infrastructure's Box2D shim, i.e. the reference's
    own Python env layer on the restated dynamics of oracle/multiwalker_ref.c: the only form in which this module runs in an image without
    Box2D.  (With pybox2d the C++ library would take the place of that C file; the Python side, which dominates, is the reference's.)"""
    seed, steps = arg
    import numpy as np
    from oracle import ref_loader
    ref_loader.load()
    shim = os.path.join(ROOT, "oracle", "shims_box2d")
    if shim not in sys.path:
        sys.path.insert(0, shim)
    os.environ.setdefault("MPLBACKEND", "Agg")
    from madrl_environments.walker.multi_walker import MultiWalkerEnv
    np.random.seed(seed)
    env = MultiWalkerEnv(n_walkers=3)
    env.seed(seed)
    rng = np.random.RandomState(seed)
    env.reset()
    for _ in range(10):
        env.step(rng.uniform(-1, 1, (3, 4)))
    env.reset()
    t_in = 0
    n_resets = 0
    t0 = time.perf_counter()
    for _ in range(steps):
        _, _, done, _ = env.step(rng.uniform(-1, 1, (3, 4)))
        t_in += 1
        if done or t_in >= 500:
            env.reset()
            t_in = 0
            n_resets += 1
    return steps, time.perf_counter() - t0, n_resets


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def measure(fn, steps, cores):
    import multiprocessing as mp
    one_steps, one_dt, one_resets = fn((0, steps))
    ctx = mp.get_context("fork")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        res = pool.map(fn, [(s, steps) for s in range(cores)])
    wall = time.perf_counter() - t0
    # aggregate = sum over processes of steps / own stepping time (imports and the pool start-up are not stepping)
    agg = sum(s / dt for s, dt, _ in res)
    return {"one_process_steps_per_s": one_steps / one_dt, "one_process_steps": one_steps, "one_process_resets": one_resets,
            "all_core_steps_per_s": agg, "processes": cores, "steps_per_process": steps,
            "slowest_process_s": max(dt for _, dt, _ in res), "pool_wall_s_incl_import": wall}


def port_same_host():
    """The C restatement (oracle/pursuit_oracle.c, OpenMP over envs, all cores) on THIS host: lets a reader carry the
    reference-to-port ratio over to the GPU box, where only the port can be timed (bench.py cpu_baseline_port).
    Runs in a child process because this one pins OMP_NUM_THREADS=1 for the NumPy reference."""
    import subprocess
    code = ("import json, sys; sys.path.insert(0, %r); import bench; from madrl_amd.maps import rectangle_map\n"
            "kw = dict(n_pursuers=8, n_evaders=30, obs_range=7, n_catch=2, surround=True, flatten=True, reward_mech='local')\n"
            "print(json.dumps(bench.cpu_baseline_port([rectangle_map(16, 16)], kw, budget_s=8.0)))" % ROOT)
    env = {k: v for k, v in os.environ.items() if not k.endswith("_NUM_THREADS")}
    r = json.loads(subprocess.check_output([sys.executable, "-c", code], env=env).decode().strip().splitlines()[-1])
    return {"pursuit_c_port_steps_per_s": r["value"], "cores": r["cores"], "sample": r["sample"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r05_cpu_reference", "record.json"))
    args = ap.parse_args()
    from oracle import ref_loader
    if not ref_loader.reference_available():
        raise SystemExit("reference tree not present (this script runs in the build container only)")
    import numpy as np
    cores = os.cpu_count()
    rec = {"what": "unmodified sisl/MADRL reference (NumPy), imported from /root/reference under oracle/shims, "
                   "one env per process, OMP_NUM_THREADS=1, random actions, reset on done or max_path_length",
           "host": {"cpu_model": cpu_model(), "logical_cores": cores, "machine": platform.machine(),
                    "where": "build container (the reference cannot travel to the GPU box)"},
           "python": platform.python_version(), "numpy": np.__version__,
           "pursuit_c1": dict(config="PursuitEvade 16x16 rectangle_map, 8 pursuers / 30 evaders, obs_range 7, n_catch 2, "
                                     "surround, flatten, local reward (BASELINE configs[0]); pursuit_evade.py:209-262",
                              **measure(run_pursuit, args.steps, cores)),
           "waterworld_c3_single_env": dict(config="MAWaterWorld(5, 10) defaults, 30 sensors; waterworld.py:220-436",
                                            **measure(run_waterworld, args.steps, cores)),
           "multiwalker_c4_single_env_over_shim": dict(
               config="MultiWalkerEnv(n_walkers=3) defaults (BASELINE configs[3]); the unmodified multi_walker.py over oracle/shims_box2d: the reference's "
                      "Python env layer on the restated dynamics (oracle/multiwalker_ref.c) -- NOT pybox2d, which this image does not have",
               **measure(run_multiwalker, max(200, args.steps // 5), cores)),
           "c_port_same_host": port_same_host(),
           "unit": "env-steps/s", "script": "scripts/cpu_reference_bench.py"}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
