#!/bin/bash
# Runs ON THE GPU BOX: GPU parity suite on the current build, then the bench A/B against libmsnake_base2.so (HEAD's kernel).
OUT=gpurun_out/r03_s23; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/gpu_tests.log 2>&1 || { tail -30 $OUT/gpu_tests.log; exit 1; }
tail -1 $OUT/gpu_tests.log
timeout -k 10 600 tools/ab_libs.sh ${1:-8} "" default $GRAFT_REPO_ROOT/self-play-on-multi-snakes-environment_amd/libmsnake_base2.so | tee $OUT/ab.txt
