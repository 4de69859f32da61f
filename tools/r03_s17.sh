#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
PKG=$R/self-play-on-multi-snakes-environment_amd
cd $R
MSNAKE_LIB=$PKG/libmsnake_ntstate3.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config3 or golden_tape" 2>&1 | tail -1
bash tools/ab_libs.sh 6 "" default $PKG/libmsnake_ntstate1.so $PKG/libmsnake_ntstate2.so $PKG/libmsnake_ntstate3.so
