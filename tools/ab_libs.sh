#!/bin/bash
# Runs ON THE GPU BOX: same-box A/B of builds of libmsnake (process-level bimodality: many alternating processes).
# usage: tools/ab_libs.sh <rounds> <bench args or ""> <lib> [<lib> ...]   ("default" = the in-tree libmsnake.so)
N=$1; shift
ARGS=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
TMP=$(mktemp)
for i in $(seq 1 $N); do
  for lib in "$@"; do
    L=$lib; [ "$lib" = default ] && L=""
    MSNAKE_LIB=$L timeout -k 10 200 python $R/bench.py --steps 1024 --warmup 64 --repeats 5 --no-cpu-baseline --no-rollout $ARGS 2>/dev/null > $TMP
    rc=$?
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT"; exit 1; fi
    python3 -c "
import json; d=json.load(open('$TMP')); print('$(basename $lib)', d['roofline']['launch_us'])"
  done
done | python3 -c "
import sys, statistics, collections
acc = collections.defaultdict(list)
for l in sys.stdin:
    k, v = l.split(); acc[k].append(float(v))
for k, v in acc.items():
    print(k, 'median', round(statistics.median(v), 3), 'min', min(v), 'mean', round(statistics.fmean(v), 3), sorted(v))
"
rm -f $TMP
