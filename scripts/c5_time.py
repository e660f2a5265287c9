"""Step time of the C5 shard (32x32, 16 v 60, 32 768 envs) through bench.py: python scripts/c5_time.py"""
import os, sys, json, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "pursuit_c5", "--steps", "200", "--warmup", "20", "--no-cpu-baseline"],
                     capture_output=True, text=True).stdout.strip().split("\n")[-1]
j = json.loads(out)
print("N=%6s  %.1f us/step  %.3e env-steps/s frac %.3f" % (j["config"]["envs_per_gpu"], j["ms_per_step"] * 1e3, j["value"], j["roofline"]["frac"]), flush=True)
