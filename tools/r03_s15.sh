#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
PKG=$R/self-play-on-multi-snakes-environment_amd
cd $R
MSNAKE_LIB=$PKG/libmsnake_flat.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused or warpframe" 2>&1 | tail -2
echo "== flat fused (real nt)"
MSNAKE_LIB=$PKG/libmsnake_flat.so timeout -k 10 300 python tools/fused_ab.py 4096 8192 32768 2>&1 | grep '^{'
echo "== dword stores (shipped)"
timeout -k 10 300 python tools/fused_ab.py 4096 8192 32768 2>&1 | grep '^{' | grep -v stream
