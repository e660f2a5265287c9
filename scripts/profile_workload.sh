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
python bench.py --workload $WL --steps 200 --warmup 20 --no-cpu-baseline "$@" 2>/dev/null | tail -1 > $OUT/bench_line.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python bench.py --workload $WL --steps 200 --warmup 20 --no-cpu-baseline "$@" > $OUT/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o p -- python bench.py --workload $WL --steps 10 --warmup 3 --no-cpu-baseline "$@" > $OUT/pmc_$c.log 2>&1
done
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
python - "$OUT" "$WL" "$KSUB" <<'PY'
import csv, glob, json, os, sys
out, wl, ksub = sys.argv[1:4]
line = json.loads(open(os.path.join(out, "bench_line.json")).read())
vals, kname = {}, None
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = []
    for f in glob.glob(os.path.join(out, "pmc_" + c, "**", "*counter_collection*.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if ksub in k and row["Counter_Name"] == c and ", 0, " not in k.split("(")[0][-14:]:
                acc.append((k, float(row["Counter_Value"])))
    # step launches only: the reset launch (MODE 0) of the same template is dropped by taking the most frequent name
    names = {}
    for k, v in acc:
        names.setdefault(k, []).append(v)
    if names and wl == "multiwalker":
        # one step = several launches of the phase kernels (plus the near-empty second pass): everything they move, per step() call
        grids = []
        for f in glob.glob(os.path.join(out, "pmc_" + c, "**", "*counter_collection*.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if ksub in row["Kernel_Name"] and row["Counter_Name"] == c and ("mw_step_kernel<1>" in row["Kernel_Name"] or "ILi1E" in row["Kernel_Name"]):
                    grids.append(int(row["Grid_Size"]))
        n_steps = sum(1 for g in grids if g == max(grids))   # the collide launch of the main pass: once per step() call
        kname = "mw_step_kernel<collide | solve | continuous pass>, all launches of a step() call"
        vals[c] = sum(v for _, v in acc) / max(n_steps, 1)
    elif names:
        kname = max(names, key=lambda k: len(names[k]))
        vals[c] = sum(names[kname]) / len(names[kname])
if len(vals) == 2:
    per_launch = line["config"].get("envs_per_launch", line["config"]["envs_per_gpu"])
    j = dict(workload=wl, kernel=kname, envs=per_launch, envs_per_gpu=line["config"]["envs_per_gpu"], streams=line["config"].get("streams_per_gpu", 1),
             FETCH_SIZE_KiB=vals["FETCH_SIZE"], WRITE_SIZE_KiB=vals["WRITE_SIZE"],
             traffic_bytes_per_launch=(vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024,
             algorithmic_bytes_per_launch=line["roofline"]["algorithmic_bytes_per_env_step"] * per_launch,
             note="separate rocprofv3 --pmc passes (scripts/profile_workload.sh); counters in KiB; every read of these kernels is <= 4 B per lane, "
                  "so the gfx950 x2 correction for wide (16 B/lane) reads does not apply")
    j["traffic_over_algorithmic"] = j["traffic_bytes_per_launch"] / j["algorithmic_bytes_per_launch"]
    json.dump(j, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(j))
else:
    print("pmc incomplete", vals)
for row in csv.DictReader(open(os.path.join(out, "kernel_stats.csv"))):
    if ksub in row["Name"]:
        print("%s calls=%s avg=%.1f us min=%.1f max=%.1f" % (row["Name"][:90], row["Calls"], float(row["AverageNs"]) / 1e3, float(row["MinNs"]) / 1e3, float(row["MaxNs"]) / 1e3))
print("bench line: value %.4g  kernel_ms %.4f  frac %.3f" % (line["value"], line["roofline"]["kernel_ms"], line["roofline"]["frac"]))
PY
rm -rf $OUT/trace $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
