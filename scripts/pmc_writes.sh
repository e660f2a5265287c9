#!/bin/bash
# L2 -> memory write requests of the Pursuit step kernel by size (64 B vs 32 B) and their stalls: counter evidence for the
# cost of partially written sectors (quirk Q2).  One --pmc pass, no trace domains.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/pmc_writes
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_WRREQ_WRITE_DRAM_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum" "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum"; do
  i=$((i+1))   # the TCC block collects two of these per pass ("exceeds the capabilities of the hardware" otherwise)
  timeout 60 rocprofv3 --pmc $set --output-format csv -d $OUT/s$i -o p -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/s$i.log 2>&1
done
echo "rc=$?"
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_writes/*/*counter_collection*.csv")):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "pursuit_wave_kernel<" in row["Kernel_Name"] and ", 1, false" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print("%-40s n=%3d mean=%.6g  per_env=%.2f" % (k, len(v), sum(v)/len(v), sum(v)/len(v)/65536))
PY
find $OUT -name "*.csv" -size +2M -delete
grep -h "exceeds" $OUT/*.log | head -3
