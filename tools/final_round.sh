#!/bin/bash
# Runs ON THE GPU BOX: everything the round's evidence under profiles/ comes from, on ONE box:
# parity suite, bench at the driver's step count and at 1 024 steps, rocprofv3 stats + PMC passes at
# 4 096 and 262 144 envs, the persistent-tape profile, the SURVEY 8(d) sweep and the 2-rank rehearsal.
# usage: tools/final_round.sh <tag>
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
bash tools/gpu_check.sh $TAG || exit 1
bash tools/profile_round.sh ${TAG}_4096 4096 1024 > $OUT/prof4096.log 2>&1
echo "profile 4096 done"
bash tools/profile_round.sh ${TAG}_32768 32768 256 > $OUT/prof32768.log 2>&1
echo "profile 32768 done"
bash tools/profile_round.sh ${TAG}_262144 262144 64 > $OUT/prof262144.log 2>&1
echo "profile 262144 done"
bash tools/profile_round.sh ${TAG}_fused4_4096 4096 512 4 > $OUT/prof_fused4_4096.log 2>&1
bash tools/profile_round.sh ${TAG}_fused4_32768 32768 64 4 > $OUT/prof_fused4_32768.log 2>&1
echo "profiles fused x4 done"
bash tools/profile_rollout.sh $TAG > $OUT/prof_rollout.log 2>&1
echo "profile rollout done"
python tools/sweep.py > $OUT/sweep.json 2> $OUT/sweep.err
echo "sweep done"
PKG=$R/self-play-on-multi-snakes-environment_amd
# what a launch cadence is made of: wave stamps of the diagnostic build, the production library's HIP-event figure and
# the rocprofv3 kernel duration, all on this box
MSNAKE_LIB=$PKG/libmsnake_dbg.so timeout -k 10 300 python tools/span_gap.py 4096 512 > $OUT/span_gap_dbg.json 2> $OUT/span_gap_dbg.err
MSNAKE_LIB=$PKG/libmsnake_span.so timeout -k 10 300 python tools/span_gap.py 4096 512 > $OUT/span_gap_light.json 2> $OUT/span_gap_light.err
echo "span/gap done"
# the collective path on RCCL with the one rank a one-GPU box can give it
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --backend nccl --steps 1024 --warmup 64 --no-cpu-baseline 2> $OUT/bench_1rank_rccl.err | grep "^{\"metric\"" > $OUT/bench_1rank_rccl.json
cut -c1-200 $OUT/bench_1rank_rccl.json
MSNAKE_BENCH_ONE_DEVICE=1 timeout -k 10 300 python bench.py --gpus 2 --backend gloo --steps 1024 --warmup 64 2> $OUT/bench_2rank.err | grep "^{\"metric\"" > $OUT/bench_2rank_gloo_one_device.json
cut -c1-300 $OUT/bench_2rank_gloo_one_device.json
