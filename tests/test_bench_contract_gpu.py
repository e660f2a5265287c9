"""GPU tests (-m gpu): bench.py keeps its output contract (one JSON line, the agreed keys, roofline object), for one
process and for the N > 1 path (two ranks on the one visible GPU over gloo: the RCCL path needs one GPU per rank)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
        "config", "roofline"}


def _last_json(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_single_process_line():
    r = subprocess.run([sys.executable, "bench.py", "--steps", "30", "--warmup", "5", "--envs", "8192", "--no-cpu-baseline"], cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _last_json(r.stdout)
    assert KEYS <= set(j) and j["n_gpus"] == 1 and j["steps"] == 30 and j["warmup"] == 5 and j["unit"] == "env-steps/s"
    assert j["value"] > 1e7 and j["higher_is_better"] is True and j["scaling"] == "weak" and j["vs_baseline"] is None
    rf = j["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4 and "workload" in j["config"]   # (six significant digits in the line)


def test_two_ranks_gather_path():
    env = dict(os.environ, MADRL_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", "bench.py", "--gpus", "2", "--steps", "70", "--warmup", "5", "--envs", "2048",
                        "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=400, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    j = _last_json(r.stdout)
    assert j["n_gpus"] == 2 and j["steps"] == 70 and j["config"]["parallelism"] == "env-sharded x2" and "cpu_baseline" not in j
    assert j["config"]["rccl_ranks"] == 2 and j["config"]["envs_total"] == 4096 and len(j["config"]["region_ms_per_step"]) == 3


def test_plain_python_gpus_2_launches_its_own_ranks():
    """`python3 bench.py --gpus 2` with no launcher around it (how the driver invokes the 1-GPU bench) must start its own two
    ranks instead of exiting: the same code path the RCCL run takes, here over gloo on the one visible GPU"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MADRL_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--envs", "2048", "--steps", "70", "--warmup", "5"], cwd=ROOT,
                       capture_output=True, text=True, timeout=400, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    j = _last_json(r.stdout)
    assert KEYS <= set(j) and j["n_gpus"] == 2 and j["steps"] == 70 and j["config"]["rccl_ranks"] == 2
    assert j["config"]["collective_backend"] == "gloo" and j["value"] > 1e5


def test_one_rank_takes_the_rccl_path():
    """backend "nccl" (= RCCL) with ONE rank on the one visible GPU: process group on the device, barrier, max over ranks, and the
    chunked asynchronous all-gather of the compact trajectory inside the timed region -- the code an 8-GPU run takes, minus the peers
    (two ranks on one device are refused by RCCL, so the two-rank tests above run over gloo)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "MADRL_BENCH_BACKEND")}
    env.update(MADRL_BENCH_FORCE_COLLECTIVE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "bench.py", "--envs", "8192", "--steps", "120", "--warmup", "5", "--prep", "50"], cwd=ROOT,
                       capture_output=True, text=True, timeout=400, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    j = _last_json(r.stdout)
    assert KEYS <= set(j) and j["n_gpus"] == 1 and j["steps"] == 120 and j["config"]["rccl_ranks"] == 1
    assert j["config"]["collective_backend"] == "nccl" and j["config"]["trajectory_gather_in_timed_region"] is True
    assert j["value"] > 1e7 and "workloads" not in j and "cpu_baseline" not in j


def test_default_batch_line_carries_every_baseline_config():
    """the default invocation (BASELINE batch sizes) times Waterworld, MultiWalker and the configs[4] shard in the same run and
    reports them under `workloads`; every launch of the headline carries fused resets (steady-state episode ages)"""
    env = dict(os.environ, MADRL_BENCH_CPU_BUDGET="0.5")   # every workload still runs its CPU sample, just a short one
    full = os.path.join(ROOT, "gpurun_out", "bench_full_test.json")
    os.makedirs(os.path.dirname(full), exist_ok=True)
    r = subprocess.run([sys.executable, "bench.py", "--steps", "8", "--warmup", "2", "--full-record", full], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][0]
    # the driver keeps the TAIL of the output: the whole line -- every BASELINE config -- must fit a few KB (round 4's 15 KB line lost
    # Waterworld and half of MultiWalker)
    assert len(line) <= 6144, len(line)
    j = _last_json(r.stdout)
    assert KEYS <= set(j) and j["config"]["envs_per_gpu"] == 65536 and j["data"] == "synthetic"
    wl = j["workloads"]
    assert set(wl) >= {"waterworld", "multiwalker", "pursuit_c5", "pursuit_colocate", "waterworld_std", "multiwalker_w10", "pursuit_rollout", "hostage",
                       "pursuit_authors", "pursuit_collective"}
    assert j["cpu_baseline"]["value"] > 0 and j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["cores"] >= 1
    rf = j["roofline"]
    # `frac` is priced by the wall clock of the timed region (ms_per_step), `frac_kernel` by the HIP events around the launches
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4 and rf["frac_kernel"] >= rf["frac"] * 0.98 and rf["one_launch_ms"] > 0
    assert len(j["config"]["region_ms_per_step"]) == 3 and min(j["config"]["region_ms_per_step"]) <= j["ms_per_step"] + 1e-6
    assert j["config"]["rccl_ranks"] == 1 and j["config"]["collective_backend"] is None
    for name, w in wl.items():
        assert "error" not in w, (name, w)
        assert w["value"] > 1e5 and w["roofline"]["frac"] > 0 and w["workload"], name
        if name == "pursuit_collective":   # the N > 1 code path in a one-rank group: recording + RCCL exchange in the timed region
            assert w["gather"] == "root" and 0.9 < w["over_headline"] < 3.0, w
        else:
            cb = w["cpu_baseline"]
            assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] == "port", name
    assert wl["pursuit_colocate"]["roofline"]["kernel"].startswith("pursuit_wave_kernel")
    assert wl["multiwalker"]["roofline"]["valu_frac"] > 0 and wl["multiwalker"]["envs"] == 16384
    assert wl["multiwalker_w10"]["roofline"]["valu_frac"] > 0 and "n_walkers=10" in wl["multiwalker_w10"]["workload"]
    # the long record (prose, every key of the earlier rounds) goes to the file
    fj = json.load(open(full))
    assert 0 < fj["config"]["horizon_resets_per_env_in_timed_region"] < 1 and fj["config"]["horizon_resets_per_step"] > 100
    assert "multiwalker_ref.c" in fj["workloads"]["multiwalker"]["cpu_baseline"]["sample"]     # the independent restatement, not the product source
    assert fj["roofline"]["frac_vs_measured_copy"] > fj["roofline"]["frac"]
