#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel trace + the two PMC passes the MI355X guide
# prescribes for HBM traffic (FETCH_SIZE and WRITE_SIZE cannot share a pass), all on the default
# bench.py command.  Output under gpurun_out/prof_<tag>/ ; tools/summarize_profile.py turns it into
# the small files committed under profiles/.
set -u
# usage: tools/profile_round.sh <tag> [envs-per-gpu] [steps] [obs-scale]   (defaults: the bench workload, 4 096 envs, 1 024 steps, native)
TAG=${1:-r01}
ENVS=${2:-4096}
STEPS=${3:-1024}
SCALE=${4:-1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="--steps $STEPS --warmup 64 --repeats 1 --envs-per-gpu $ENVS --obs-scale $SCALE --no-cpu-baseline --no-rollout"
# the same command without the profiler, on the same box: rocprofv3 serialises the dispatches and
# lengthens both the kernels and the gaps between them, and boxes differ by a few per cent
python3 $R/bench.py $ARGS > $OUT/bench_plain.json 2> $OUT/bench_plain.err || echo "plain bench failed"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py $ARGS > $OUT/kt.log 2>&1 || echo "kernel-trace pass failed"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1 || echo "FETCH_SIZE pass failed"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $ARGS > $OUT/pmc_write.log 2>&1 || echo "WRITE_SIZE pass failed"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d $OUT/pmc_insts -- python3 $R/bench.py $ARGS > $OUT/pmc_insts.log 2>&1 || echo "insts pass failed"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_cycles -- python3 $R/bench.py $ARGS > $OUT/pmc_cycles.log 2>&1 || echo "cycles pass failed"
grep -h '"metric"' $OUT/*.log | head -5
find $OUT -name "*.csv" | head -20
