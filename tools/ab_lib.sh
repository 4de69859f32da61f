#!/bin/bash
# same-box A/B of two builds of libmsnake.so: tools/ab_lib.sh <other.so> [envs ...]
OTHER=$1; shift
ENVS=${@:-"4096 32768 262144"}
for i in 1 2; do
  for lib in "" "$OTHER"; do
    echo "== lib=${lib:-current}"
    MSNAKE_LIB=$lib python tools/kbench.py --envs $ENVS --iters 200 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('  envs', d['envs'], 'step', d['step_us'], 'noobs', d['step_noobs_us'], 'rollout', d.get('rollout_step_us'), 'reset', d['reset_us'])"
  done
done
