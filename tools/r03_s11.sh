#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
TAG=r03a
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
bash tools/profile_round.sh ${TAG}_32768 32768 256 > $OUT/prof32768.log 2>&1
bash tools/profile_round.sh ${TAG}_fused4_32768 32768 64 4 > $OUT/prof_fused4_32768.log 2>&1
python3 -c "
import json
for t in ('32768','fused4_32768'):
    b=json.load(open('gpurun_out/prof_${TAG}_'+t+'/bench_plain.json')); print(t,'plain same box', b['roofline']['launch_us'], b['roofline']['frac'])"
