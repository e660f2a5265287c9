#!/bin/bash
# rocprofv3 kernel stats for the Waterworld and MultiWalker workloads
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/prof_other
rm -rf $OUT; mkdir -p $OUT
for wl in ${WORKLOADS:-waterworld multiwalker hostage}; do
  python bench.py --workload $wl --steps 100 --warmup 10 > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err
  tail -1 $OUT/bench_$wl.json | cut -c1-600
  timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$wl -o $wl -- python bench.py --workload $wl --steps 50 --warmup 5 --no-cpu-baseline > $OUT/trace_$wl.log 2>&1
  head -3 $OUT/$wl/*kernel_stats.csv | cut -c1-260
done
find $OUT -name "*.csv" -size +2M -delete
