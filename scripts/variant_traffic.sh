#!/bin/bash
# HBM write / fetch traffic and kernel time of profiling variants of one workload (scripts/variants.sh builds them).
#   scripts/variant_traffic.sh <workload> <kernel-substring> <src> <variant> [<variant> ...]     ("base" = the shipped library)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
WL=$1; KSUB=$2; SRC=$3; shift 3
for v in "$@"; do
  if [ "$v" = base ]; then unset MADRL_HIP_LIB; else export MADRL_HIP_LIB=$PWD/scripts/_variants/libmadrl_hip.$SRC.$v.so; fi
  OUT=gpurun_out/variant_traffic/$WL.$v; rm -rf $OUT; mkdir -p $OUT
  LINE=$(python bench.py --workload $WL --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1)
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $c --output-format csv -d $OUT/$c -o p -- python bench.py --workload $WL --steps 10 --warmup 3 --no-cpu-baseline > $OUT/$c.log 2>&1
  done
  python - "$OUT" "$KSUB" "$v" "$LINE" <<'PY'
import csv, glob, json, sys, collections
out, ksub, v, line = sys.argv[1:5]
j = json.loads(line)
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    names = collections.defaultdict(list)
    for f in glob.glob(out + "/" + c + "/**/*counter_collection*.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if ksub in row["Kernel_Name"] and row["Counter_Name"] == c:
                names[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    if names:
        k = max(names, key=lambda n: len(names[n]))
        res[c] = sum(names[k]) / len(names[k]) * 1024
n = j["config"]["envs_per_gpu"]
print("variant %-5s kernel %.1f us   fetch %.0f B/env   write %.0f B/env   (algorithmic %d B/env)" % (
    v, j["roofline"]["kernel_ms"] * 1e3, res.get("FETCH_SIZE", float("nan")) / n, res.get("WRITE_SIZE", float("nan")) / n,
    j["roofline"]["algorithmic_bytes_per_env_step"]))
PY
  rm -rf $OUT
done
