#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/pmc_insts
rm -rf $OUT; mkdir -p $OUT
timeout 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --output-format csv -d $OUT/a -o p -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/a.log 2>&1
echo "rc=$?"
timeout 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -o p -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/b.log 2>&1
echo "rc=$?"
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_insts/*/*counter_collection*.csv")):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "pursuit" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("==", f)
    for k, v in sorted(acc.items()):
        print("%-24s n=%3d mean=%.6g  per_env=%.2f" % (k, len(v), sum(v)/len(v), sum(v)/len(v)/65536))
PY
find $OUT -name "*.csv" -size +2M -delete
