#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for pol in stream plain; do
  for i in 1 2 3; do
    timeout -k 10 300 python bench.py --steps 1024 --warmup 64 --repeats 5 --no-cpu-baseline --legs strided --store-policy $pol 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('$pol', 'reused buffer', d['roofline']['launch_us'], 'strided', d['per_step_strided']['us_per_step'])"
  done
done
