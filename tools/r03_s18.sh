#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
PKG=$R/self-play-on-multi-snakes-environment_amd
cd $R
mkdir -p gpurun_out/r03q
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03q/pytest.log 2>&1; rc=$?; tail -2 gpurun_out/r03q/pytest.log
[ $rc -ne 0 ] && { grep -E "Error|assert |FAILED" gpurun_out/r03q/pytest.log | head; exit 1; }
bash tools/pmc_insts.sh r03q/insts 4096 2>&1 | tail -2
echo "== A/B at 4096, 10 rounds"
bash tools/ab_libs.sh 10 "" default $PKG/libmsnake_prev.so
show() { grep '^{' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  envs', d['envs'], 'step', d['step_us'], 'frac', round(d['alg_GBs']/8000, 3))"; }
for lib in default prev; do
  L=$PKG/libmsnake_$lib.so; [ $lib = default ] && L=""
  echo "lib=$lib"; MSNAKE_LIB=$L timeout -k 10 300 python tools/kbench.py --envs 8192 32768 262144 --iters 150 2>/dev/null | show
done
