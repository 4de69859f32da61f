#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r03p
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03p/pytest.log 2>&1; rc=$?; tail -3 gpurun_out/r03p/pytest.log
[ $rc -ne 0 ] && { grep -E "Error|assert |FAILED" gpurun_out/r03p/pytest.log | head; exit 1; }
timeout -k 10 300 python tools/fused_ab.py 4096 8192 32768 2>&1 | grep '^{' | tee gpurun_out/r03p/fused.txt
