#!/bin/bash
# Which unit is busy?  Separate --pmc passes (no trace domains), summary under gpurun_out/pmc_units.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/pmc_units
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES SQ_LDS_BANK_CONFLICT" \
           "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum GRBM_GUI_ACTIVE" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_WRITE_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --output-format csv -d $OUT/s$i -o p -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/s$i.log 2>&1
  echo "set $i rc=$?"
done
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_units/*/*counter_collection*.csv")):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "pursuit_wave" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("==", f)
    for k, v in sorted(acc.items()):
        print("%-32s n=%3d mean=%.6g  per_env=%.2f" % (k, len(v), sum(v)/len(v), sum(v)/len(v)/65536))
PY
find $OUT -name "*.csv" -size +2M -delete
tail -3 $OUT/s*.log | grep -i -E "error|invalid|not found" | head
