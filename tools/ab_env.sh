#!/bin/bash
# same-box A/B of environment knobs of the library: tools/ab_env.sh "VAR=a" "VAR=b" ... (bench at 4 096 envs, 3 rounds)
mkdir -p gpurun_out/ab_env
for i in $(seq 1 ${AB_ROUNDS:-3}); do
  for kv in "$@"; do
    env $kv python bench.py --steps 1024 --warmup 64 --repeats 5 --no-cpu-baseline --no-rollout > gpurun_out/ab_env/b.json 2>/dev/null
    python -c "
import json
d=json.load(open('gpurun_out/ab_env/b.json')); print('$kv', d['roofline']['launch_us'])"
  done
done
