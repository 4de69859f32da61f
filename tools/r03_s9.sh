#!/bin/bash
set -u
TAG=${1:-r03i}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
PKG=$R/self-play-on-multi-snakes-environment_amd
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; tail -4 $OUT/pytest.log
[ $rc -ne 0 ] && { grep -E "Error|assert |FAILED" $OUT/pytest.log | head -20; exit 1; }
echo "== A/B headline: this build vs the last commit, 8 rounds"
bash tools/ab_libs.sh 8 "" default $PKG/libmsnake_prev.so
