#!/bin/bash
# Interleaved A/B timing of library variants at the headline shape: scripts/ab.sh <rounds> <lib|-> ...   ("-" = the shipped library)
# Each round runs every variant once (fresh process, scripts/size_sweep.py 65536); prints min / median per variant.
cd $GRAFT_REPO_ROOT 2>/dev/null || cd "$(dirname "$0")/.."
R=$1; shift
SIZES=${SIZES:-65536}
for r in $(seq $R); do
  for v in "$@"; do
    if [ "$v" = "-" ]; then L=""; else L="$PWD/scripts/_variants/libmadrl_hip.$v.so"; fi
    MADRL_HIP_LIB=$L python ${SCRIPT:-scripts/size_sweep.py} $SIZES 2>/dev/null | grep "N=" | awk -v v="$v" '{print v, $2, $3}'
  done
done | python -c "
import sys, collections, statistics
d = collections.defaultdict(list)
for l in sys.stdin:
    v, n, t = l.split()[:3]
    d[(v, n)].append(float(t))
for k, ts in sorted(d.items()):
    print('%-14s N=%-7s min %.1f  median %.1f  max %.1f us  (%d runs)' % (k[0], k[1], min(ts), statistics.median(ts), max(ts), len(ts)))
"
