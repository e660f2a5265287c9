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
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and "workload" in j["config"]


def test_two_ranks_gather_path():
    env = dict(os.environ, MADRL_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", "bench.py", "--gpus", "2", "--steps", "70", "--warmup", "5", "--envs", "2048",
                        "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=400, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    j = _last_json(r.stdout)
    assert j["n_gpus"] == 2 and j["steps"] == 70 and j["config"]["parallelism"] == "env-sharded x2" and "cpu_baseline" not in j


def test_default_batch_line_carries_every_baseline_config():
    """the default invocation (BASELINE batch sizes) times Waterworld, MultiWalker and the configs[4] shard in the same run and
    reports them under `workloads`; every launch of the headline carries fused resets (steady-state episode ages)"""
    r = subprocess.run([sys.executable, "bench.py", "--steps", "8", "--warmup", "2", "--no-cpu-baseline"], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _last_json(r.stdout)
    assert KEYS <= set(j) and j["config"]["envs_per_gpu"] == 65536
    assert 0 < j["config"]["horizon_resets_per_env_in_timed_region"] < 1 and j["config"]["horizon_resets_per_launch"] > 100
    wl = j["workloads"]
    assert set(wl) == {"waterworld", "multiwalker", "pursuit_c5"}
    for name, w in wl.items():
        assert "error" not in w, (name, w)
        assert w["value"] > 1e5 and w["roofline"]["frac"] > 0 and "workload" in w["config"], name
    assert wl["multiwalker"]["roofline"]["valu_frac"] > 0 and wl["multiwalker"]["config"]["envs_per_gpu"] == 16384
