#!/bin/bash
# HBM traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, no trace domains) of the step kernels of the other
# workloads -> gpurun_out/pmc_other/<workload>/pmc_traffic.json (copy into profiles/<dir>/ to have bench.py report it)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/pmc_other
rm -rf $OUT; mkdir -p $OUT
for wl in ${WORKLOADS:-waterworld hostage multiwalker}; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 120 rocprofv3 --pmc $c --output-format csv -d $OUT/$wl/$c -o p -- python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline > $OUT/$wl.$c.log 2>&1
  done
done
python - <<'PY'
import csv, glob, json, os, collections
envs = dict(waterworld=32768, hostage=32768, multiwalker=16384)
for wl in envs:
    vals = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        acc = collections.defaultdict(list)
        for f in glob.glob("gpurun_out/pmc_other/%s/%s/**/*counter_collection*.csv" % (wl, c), recursive=True):
            for row in csv.DictReader(open(f)):
                if wl in row["Kernel_Name"] and "kernel<1" in row["Kernel_Name"] and row["Counter_Name"] == c:
                    acc[c].append(float(row["Counter_Value"]))
        if acc[c]:
            vals[c] = sum(acc[c]) / len(acc[c])
    if len(vals) == 2:
        j = dict(workload=wl, kernel="%s_kernel<1,...>" % wl, envs=envs[wl], FETCH_SIZE_KiB=vals["FETCH_SIZE"], WRITE_SIZE_KiB=vals["WRITE_SIZE"],
                 traffic_bytes_per_launch=(vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024,
                 note="separate rocprofv3 --pmc passes (scripts/pmc_traffic_other.sh); counters in KiB; reads are <= 4 B per lane, no wide-read correction")
        os.makedirs("gpurun_out/pmc_other/%s" % wl, exist_ok=True)
        json.dump(j, open("gpurun_out/pmc_other/%s/pmc_traffic.json" % wl, "w"), indent=1)
        print(wl, json.dumps(j)[:300])
    else:
        print(wl, "incomplete", vals)
PY
find $OUT -name "*.csv" -size +2M -delete
