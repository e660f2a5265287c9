"""Step time of the hostage kernel (ContinuousHostageWorld(3, 10, 5), 32 768 envs): python scripts/hw_time.py [n_envs]"""
import os, sys, json, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for N in (sys.argv[1:] or ["32768"]):
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "hostage", "--envs", N, "--steps", "300", "--warmup", "30",
                          "--no-cpu-baseline"], capture_output=True, text=True).stdout.strip().split("\n")[-1]
    j = json.loads(out)
    print("N=%6s  %.1f us/step  %.3e env-steps/s" % (N, j["ms_per_step"] * 1e3, j["value"]), flush=True)
