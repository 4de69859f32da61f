#!/bin/bash
set -u
TAG=${1:-r03h}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
PKG=$R/self-play-on-multi-snakes-environment_amd
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; tail -4 $OUT/pytest.log
[ $rc -ne 0 ] && { grep -E "Error|assert |FAILED" $OUT/pytest.log | head -20; exit 1; }
show() { grep '^{' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  envs', d['envs'], 'step', d['step_us'], 'alg_GBs', d['alg_GBs'], 'frac', round(d['alg_GBs']/8000, 3), 'rollout', d.get('rollout_step_us'), 'reset', d['reset_us'])"; }
for i in 1 2; do
for lib in default prev; do
  L=$PKG/libmsnake_$lib.so; [ $lib = default ] && L=""
  echo "lib=$lib"
  MSNAKE_LIB=$L timeout -k 10 300 python tools/kbench.py --envs 4096 8192 16384 32768 65536 262144 --iters 150 2>/dev/null | show
done
done
echo "8192 envs: plain (aligned) vs stream"
timeout -k 10 300 python tools/kbench.py --envs 4096 8192 --iters 300 --store-policy plain 2>/dev/null | show
timeout -k 10 300 python tools/kbench.py --envs 4096 8192 --iters 300 --store-policy stream 2>/dev/null | show
