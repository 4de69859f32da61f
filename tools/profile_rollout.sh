#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 evidence for the persistent msnake_rollout_tape kernel (MODE 3):
# kernel trace + FETCH_SIZE / WRITE_SIZE passes of bench.py WITH its rollout leg (256 steps per launch).
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_${TAG}_rollout
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 1024 --warmup 256 --repeats 1 --no-cpu-baseline --legs tape"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py $ARGS > $OUT/kt.log 2>&1 || echo "kernel-trace pass failed"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1 || echo "FETCH pass failed"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $ARGS > $OUT/pmc_write.log 2>&1 || echo "WRITE pass failed"
grep -h '"metric"' $OUT/kt.log | cut -c1-200
