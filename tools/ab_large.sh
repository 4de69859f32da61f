#!/bin/bash
# Runs ON THE GPU BOX: aligned vs byte-aligned copy-out at bandwidth-bound batch sizes, alternating fresh processes
# (where a 1 GB observation tensor lands physically moves a launch by up to 10 %: only many processes tell).
# usage: tools/ab_large.sh <rounds> <envs ...>
N=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
PKG=$R/self-play-on-multi-snakes-environment_amd
for i in $(seq 1 $N); do
  for lib in default noalign; do
    L=$PKG/libmsnake_$lib.so; [ $lib = default ] && L=""
    MSNAKE_LIB=$L timeout -k 10 300 python $R/tools/kbench.py --envs "$@" --iters 100 2>/dev/null | grep '^{' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$lib', d['envs'], d['step_us'])"
  done
done | python3 -c "
import sys, statistics, collections
acc = collections.defaultdict(list)
for l in sys.stdin:
    k, n, v = l.split(); acc[(int(n), k)].append(float(v))
for (n, k), v in sorted(acc.items()):
    print(n, k, 'median', round(statistics.median(v), 2), 'min', min(v), 'max', max(v), sorted(v))
"
