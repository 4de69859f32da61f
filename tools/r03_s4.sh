#!/bin/bash
set -u
TAG=${1:-r03d}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
PKG=$R/self-play-on-multi-snakes-environment_amd
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused or warpframe or obs_alignment" > $OUT/pytest_fused.log 2>&1; tail -3 $OUT/pytest_fused.log
echo "== fused flat copy-out (this build)"
timeout -k 10 300 python tools/fused_ab.py 4096 32768 2>&1 | tee $OUT/fused_new.txt
echo "== previous build (dword stores)"
MSNAKE_LIB=$PKG/libmsnake_prev.so timeout -k 10 300 python tools/fused_ab.py 4096 32768 2>&1 | grep -v stream | tee $OUT/fused_prev.txt
