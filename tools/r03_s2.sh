#!/bin/bash
set -u
TAG=${1:-r03b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
PKG=$R/self-play-on-multi-snakes-environment_amd
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
echo "== A/B late refill, 10 rounds"
bash tools/ab_libs.sh 10 "" default $PKG/libmsnake_nolate.so | tee $OUT/ab_late.txt
echo "== span/gap: late vs nolate (dbg builds)"
for v in dbg dbg_nolate; do
  MSNAKE_LIB=$PKG/libmsnake_$v.so timeout -k 10 300 python tools/span_gap.py 4096 512 > $OUT/span_gap_$v.json 2> $OUT/span_gap_$v.err || tail -3 $OUT/span_gap_$v.err
  python3 -c "
import json; d=json.load(open('$OUT/span_gap_$v.json')); print('$v', d['summary']); r=d['regions'][2]; [print(' ', k, r[k]) for k in ('wave_life_us','wave_stages_us_median','by_class','last_32_finishers_per_launch','last_ack_per_xcd_us_since_launch_start','first_to_last_wave_start_us')]"
done
