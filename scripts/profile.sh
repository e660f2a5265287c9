#!/bin/bash
# rocprofv3 passes for the bench workload; summaries get copied into profiles/ by hand.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 200 --warmup 20 --no-cpu-baseline"
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o pursuit -- python bench.py $ARGS > $OUT/trace.log 2>&1
echo "trace rc=$?"
timeout 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pursuit -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
echo "pmc fetch rc=$?"
timeout 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pursuit -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
echo "pmc write rc=$?"
find $OUT -type f | head -40
for f in $(find $OUT/trace -name "*stats*csv"); do echo "== $f"; head -20 $f; done
python scripts/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1; cat $OUT/pmc_summary.txt
# keep the merge small
find $OUT -name "*.csv" -size +4M -delete
