#!/bin/bash
# One workload, everything the judge reads: rocprofv3 kernel trace + stats, separate --pmc FETCH_SIZE / WRITE_SIZE passes
# (never combined with trace domains), the un-profiled bench line.  Result: gpurun_out/profile/<name>/ with
# kernel_stats.csv, pmc_traffic.json, bench_line.json -> copy to profiles/<round>_<name>/.
#   scripts/profile_workload.sh <workload> <name> <kernel-substring> [extra bench args]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
WL=$1; NAME=$2; KSUB=$3; shift 3
OUT=gpurun_out/profile/$NAME
rm -rf $OUT; mkdir -p $OUT
python bench.py --workload $WL --steps 200 --warmup 20 --no-cpu-baseline --no-workloads --full "$@" 2>/dev/null | tail -1 > $OUT/bench_line.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python bench.py --workload $WL --steps 200 --warmup 20 --no-cpu-baseline --no-workloads --full "$@" > $OUT/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o p -- python bench.py --workload $WL --steps 10 --warmup 3 --no-cpu-baseline --no-workloads --full "$@" > $OUT/pmc_$c.log 2>&1
done
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
python - "$OUT" "$WL" "$KSUB" <<'PY'
import csv, glob, json, os, sys
out, wl, ksub = sys.argv[1:4]
line = json.loads(open(os.path.join(out, "bench_line.json")).read())
S = line["config"].get("streams_per_gpu", 1)
per_launch = line["config"].get("envs_per_launch", line["config"]["envs_per_gpu"])
# the PMC passes ran `--steps 10 --warmup 3` (MultiWalker: its own minimum of 200 warm-up steps): every region of the bench repeats the K steps
pmc_line = None
vals, kernels = {}, set()
WIDE = ("pursuit_policy_rows_kernel",)   # kernels that read 16 bytes per lane: FETCH_SIZE under-counts those by 2 on gfx950 (MI355X_MICROARCH.md, HBM / rocprofv3 section)
wide_fetch = 0.0
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    tot = 0.0
    for f in glob.glob(os.path.join(out, "pmc_" + c, "**", "*counter_collection*.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if ksub in row["Kernel_Name"] and row["Counter_Name"] == c:
                w = 2.0 if c == "FETCH_SIZE" and any(k in row["Kernel_Name"] for k in WIDE) else 1.0
                tot += w * float(row["Counter_Value"]); kernels.add(row["Kernel_Name"].split("(")[0][:120])
                wide_fetch += float(row["Counter_Value"]) if w == 2.0 else 0.0
    try:
        pmc_line = json.loads([l for l in open(os.path.join(out, "pmc_%s.log" % c)) if l.startswith("{")][-1])
    except Exception:
        pmc_line = None
    if tot > 0 and pmc_line:
        prep = pmc_line["config"].get("prep_steps", 0)   # Pursuit: untimed steps that bring the stale-zero masks to their equilibrium
        n_steps = prep + pmc_line["warmup"] + pmc_line["config"]["timed_regions"] * pmc_line["steps"]   # step() calls of the whole batch (+ one reset launch: < 1 %)
        if S > 1 and "one_launch_per_step" in pmc_line["roofline"]:
            n_steps += prep + min(pmc_line["warmup"], 20) + pmc_line["config"]["timed_regions"] * pmc_line["steps"]   # the one-launch-per-step reference pass of the same run
        n_steps = pmc_line["config"].get("step_calls_in_process", n_steps)   # the rollout workload counts its own (warm-up in horizons, not steps)
        vals[c] = tot / n_steps
if len(vals) == 2:
    step_bytes = (vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024
    j = dict(workload=wl, kernels=sorted(kernels), envs=per_launch, envs_per_gpu=line["config"]["envs_per_gpu"], streams=S,
             FETCH_SIZE_KiB_per_step=vals["FETCH_SIZE"], WRITE_SIZE_KiB_per_step=vals["WRITE_SIZE"],
             traffic_bytes_per_step=step_bytes, traffic_bytes_per_launch=step_bytes / S,
             algorithmic_bytes_per_launch=line["roofline"]["algorithmic_bytes_per_env_step"] * per_launch,
             note="separate rocprofv3 --pmc passes (scripts/profile_workload.sh); counters in KiB, summed over every launch of the matching kernels and "
                  "divided by the step() calls of the run; reads of these kernels are <= 8 B per lane, so the gfx950 x2 correction for wide (16 B/lane) "
                  "reads does not apply, except: obsnorm_pairs_kernel (16-byte words: its FETCH_SIZE is doubled below when it is the kernel asked for) and "
                  "pursuit_policy_rows_kernel (16-byte loads: its FETCH_SIZE is counted twice in the sums above)")
    if wide_fetch:
        j["FETCH_SIZE_KiB_raw_of_16_byte_readers_in_run"] = wide_fetch
    if "obsnorm_pairs" in ksub:   # 16 B/lane loads: FETCH_SIZE under-counts by 2 on gfx950 (MI355X_MICROARCH.md, HBM / rocprofv3 section)
        j["FETCH_SIZE_KiB_per_step_corrected"] = 2 * vals["FETCH_SIZE"]
        j["traffic_bytes_per_step"] = (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024
        j["traffic_bytes_per_launch"] = j["traffic_bytes_per_step"] / S
    j["traffic_over_algorithmic"] = j["traffic_bytes_per_launch"] / j["algorithmic_bytes_per_launch"]
    json.dump(j, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(j))
else:
    print("pmc incomplete", vals)
for row in csv.DictReader(open(os.path.join(out, "kernel_stats.csv"))):
    if ksub in row["Name"]:
        print("%s calls=%s avg=%.1f us min=%.1f max=%.1f" % (row["Name"][:90], row["Calls"], float(row["AverageNs"]) / 1e3, float(row["MinNs"]) / 1e3, float(row["MaxNs"]) / 1e3))
print("bench line: value %.4g  kernel_ms %.4f  frac %.3f  streams %d" % (line["value"], line["roofline"]["kernel_ms"], line["roofline"]["frac"], S))
PY
rm -rf $OUT/trace $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
