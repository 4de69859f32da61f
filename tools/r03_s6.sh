#!/bin/bash
set -u
TAG=${1:-r03g}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
PKG=$R/self-play-on-multi-snakes-environment_amd
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
show() { grep '^{' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  envs', d['envs'], 'step', d['step_us'], 'alg_GBs', d['alg_GBs'], 'frac', round(d['alg_GBs']/8000, 3))"; }
for lib in default al1 al2 al3 al4 al5 al6; do
  L=$PKG/libmsnake_$lib.so; [ $lib = default ] && L=""
  echo "lib=$lib (auto policy: plain at 32768, streaming kind at 262144)"
  MSNAKE_LIB=$L timeout -k 10 300 python tools/kbench.py --envs 32768 262144 --iters 120 2>/dev/null | show
done
echo "aligned, plain stores at 262144 / default, plain at 262144"
MSNAKE_LIB=$PKG/libmsnake_al1.so timeout -k 10 300 python tools/kbench.py --envs 262144 --iters 120 --store-policy plain 2>/dev/null | show
timeout -k 10 300 python tools/kbench.py --envs 262144 --iters 120 --store-policy plain 2>/dev/null | show
echo "aligned nt at 32768 / default nt at 32768"
MSNAKE_LIB=$PKG/libmsnake_al1.so timeout -k 10 300 python tools/kbench.py --envs 32768 65536 131072 --iters 120 --store-policy stream 2>/dev/null | show
timeout -k 10 300 python tools/kbench.py --envs 32768 65536 131072 --iters 120 --store-policy stream 2>/dev/null | show
echo "aligned plain at 65536 131072 / default plain"
MSNAKE_LIB=$PKG/libmsnake_al1.so timeout -k 10 300 python tools/kbench.py --envs 8192 16384 65536 131072 --iters 120 --store-policy plain 2>/dev/null | show
timeout -k 10 300 python tools/kbench.py --envs 8192 16384 65536 131072 --iters 120 --store-policy plain 2>/dev/null | show
