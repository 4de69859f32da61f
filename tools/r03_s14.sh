#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
PKG=$R/self-play-on-multi-snakes-environment_amd
cd $R
mkdir -p gpurun_out/r03n
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03n/pytest.log 2>&1; rc=$?; tail -3 gpurun_out/r03n/pytest.log
[ $rc -ne 0 ] && { grep -E "Error|assert |FAILED" gpurun_out/r03n/pytest.log | head; exit 1; }
show() { grep '^{' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  envs', d['envs'], 'step', d['step_us'], 'frac', round(d['alg_GBs']/8000, 3), 'reset', d['reset_us'])"; }
echo "== store policy by batch size, aligned shape (this build)"
for pol in stream plain; do
  echo "policy $pol"
  timeout -k 10 300 python tools/kbench.py --envs 4096 8192 16384 32768 49152 65536 131072 --iters 150 --store-policy $pol 2>/dev/null | show
done
echo "== previous commit (byte-aligned nt / aligned plain), auto policy"
MSNAKE_LIB=$PKG/libmsnake_prev.so timeout -k 10 300 python tools/kbench.py --envs 4096 8192 16384 32768 49152 65536 131072 262144 --iters 150 2>/dev/null | show
echo "== this build, auto policy"
timeout -k 10 300 python tools/kbench.py --envs 4096 8192 16384 32768 49152 65536 131072 262144 --iters 150 2>/dev/null | show
echo "== A/B at 4096: this build vs previous commit, 10 rounds"
bash tools/ab_libs.sh 10 "" default $PKG/libmsnake_prev.so
