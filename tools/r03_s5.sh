#!/bin/bash
set -u
TAG=${1:-r03e}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
PKG=$R/self-play-on-multi-snakes-environment_amd
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
echo "== parity of the aligned copy-out build"
MSNAKE_LIB=$PKG/libmsnake_aligned.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ragged or config3 or large_batches_byte or golden_tape or unusual_grid or edge_case" > $OUT/pytest_aligned.log 2>&1; tail -3 $OUT/pytest_aligned.log
echo "== kbench: default vs aligned"
for i in 1 2; do
for lib in "" $PKG/libmsnake_aligned.so; do
  echo "lib=${lib:-default}"
  MSNAKE_LIB=$lib timeout -k 10 300 python tools/kbench.py --envs 4096 32768 262144 --iters 200 2>/dev/null | grep '^{' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  envs', d['envs'], 'step', d['step_us'], 'render', d['render_us'], 'alg_GBs', d['alg_GBs'])"
done
done
echo "== traffic (aligned build), 262144 envs"
MSNAKE_LIB=$PKG/libmsnake_aligned.so bash tools/pmc_traffic_split.sh $TAG/split_aligned 262144 2>&1 | tail -32
exit 0
MSNAKE_LIB=$PKG/libmsnake_span.so timeout -k 10 300 python tools/span_gap.py 4096 512 > $OUT/span_gap_light.json 2> $OUT/span_gap_light.err || tail -3 $OUT/span_gap_light.err
python3 -c "
import json; d=json.load(open('$OUT/span_gap_light.json')); print(d['summary']); r=d['regions'][2]; [print(' ', k, r[k]) for k in ('wave_life_us','by_class','first_to_last_wave_start_us','last_ack_per_xcd_us_since_launch_start')]"
MSNAKE_LIB=$PKG/libmsnake_span.so timeout -k 10 200 python bench.py --steps 1024 --warmup 64 --repeats 5 --no-cpu-baseline --no-rollout 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('span build without a span buffer:', d['roofline']['launch_us'])"
timeout -k 10 200 python bench.py --steps 1024 --warmup 64 --repeats 5 --no-cpu-baseline --no-rollout 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('production:', d['roofline']['launch_us'])"
