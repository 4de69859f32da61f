#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
PKG=$R/self-play-on-multi-snakes-environment_amd
cd $R
mkdir -p gpurun_out/r03l
export TMPDIR=/tmp
show() { grep '^{' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  envs', d['envs'], 'step', d['step_us'], 'alg_GBs', d['alg_GBs'], 'frac', round(d['alg_GBs']/8000, 3))"; }
MSNAKE_LIB=$PKG/libmsnake_halfnt.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ragged or large_batches_byte or config3" 2>&1 | tail -2
for lib in default halfnt fullnt; do
  L=$PKG/libmsnake_$lib.so; [ $lib = default ] && L=""
  echo "lib=$lib"
  MSNAKE_LIB=$L timeout -k 10 300 python tools/kbench.py --envs 4096 65536 262144 --iters 120 2>/dev/null | show
done
MSNAKE_LIB=$PKG/libmsnake_halfnt.so bash tools/pmc_traffic_split.sh r03l/split_halfnt 262144 2>&1 | grep -A3 '"step <0, 3, 0, 1>"\|"render <0, 3, 2, 1>"'
