#!/bin/bash
# Runs ON THE GPU BOX: observation store policy (plain vs streaming) by batch size, same box.
for n in 4096 8192 16384 32768 65536 131072; do
  for nt in plain stream; do
    python bench.py --store-policy $nt --steps 256 --warmup 32 --repeats 5 --envs-per-gpu $n --no-cpu-baseline --no-rollout 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('envs $n nt=$nt', d['roofline']['launch_us'], d['roofline']['frac'])"
  done
done
