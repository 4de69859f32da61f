#!/bin/bash
set -u
TAG=${1:-r03c}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
PKG=$R/self-play-on-multi-snakes-environment_amd
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
MSNAKE_LIB=$PKG/libmsnake_late12.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config3 or record_policy or draw_counter or small_boards or golden_tape" > $OUT/pytest_late12.log 2>&1; tail -2 $OUT/pytest_late12.log
MSNAKE_LIB=$PKG/libmsnake_late1.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config3 or record_policy or draw_counter or small_boards or golden_tape" > $OUT/pytest_late1.log 2>&1; tail -2 $OUT/pytest_late1.log
echo "== A/B opportunistic late refill, 8 rounds"
bash tools/ab_libs.sh 8 "" default $PKG/libmsnake_late12.so $PKG/libmsnake_late1.so | tee $OUT/ab_late.txt
for v in dbg_late12; do
  MSNAKE_LIB=$PKG/libmsnake_$v.so timeout -k 10 300 python tools/span_gap.py 4096 512 > $OUT/span_gap_$v.json 2> $OUT/span_gap_$v.err || tail -3 $OUT/span_gap_$v.err
  python3 -c "
import json; d=json.load(open('$OUT/span_gap_$v.json')); print('$v', d['summary']); r=d['regions'][2]; [print(' ', k, r[k]) for k in ('wave_life_us','wave_stages_us_median','by_class','last_32_finishers_per_launch')]"
done
