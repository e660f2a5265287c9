#!/bin/bash
# Instruction mix and issue / wait cycles of one workload's step kernel (separate --pmc passes, no trace domains).
#   scripts/pmc_mix.sh <workload> <kernel-substring>[,<kernel-substring>...] <envs> [extra bench args]   -> gpurun_out/pmc_mix/<workload>/summary.txt
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
WL=$1; KSUB=$2; ENVS=$3; shift 3
OUT=gpurun_out/pmc_mix/$WL
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_FLAT SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $OUT/s$i -o p -- python bench.py --workload $WL --steps 6 --warmup 2 --no-cpu-baseline "$@" > $OUT/s$i.log 2>&1
  echo "set $i rc=$?"
done
python - "$OUT" "$KSUB" "$ENVS" <<'PY' | tee $OUT/summary.txt
import csv, glob, collections, sys
out, ksubs, envs = sys.argv[1], sys.argv[2].split(","), float(sys.argv[3])
for ksub in ksubs:
  for f in sorted(glob.glob(out + "/*/**/*counter_collection*.csv", recursive=True)):
      acc = collections.defaultdict(list)
      names = collections.Counter()
      for row in csv.DictReader(open(f)):
          if ksub in row["Kernel_Name"]:
              names[row["Kernel_Name"]] += 1
      if not names:
          continue
      kname = max(names, key=names.get)   # the step launches, not the single reset launch
      for row in csv.DictReader(open(f)):
          if row["Kernel_Name"] == kname:
              acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
      print("==", kname[:100])
      for k, v in sorted(acc.items()):
          print("%-28s n=%3d mean=%.6g  per_env=%.1f" % (k, len(v), sum(v) / len(v), sum(v) / len(v) / envs))
PY
rm -rf $OUT/s1 $OUT/s2 $OUT/s3
