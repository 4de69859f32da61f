#!/bin/bash
# Runs ON THE GPU BOX: what the slow-path waves cost a launch (span over the fast-path waves only), light-stamp build.
OUT=gpurun_out/r03_s20; mkdir -p $OUT
PKG=$GRAFT_REPO_ROOT/self-play-on-multi-snakes-environment_amd
MSNAKE_LIB=$PKG/libmsnake_span.so timeout -k 10 300 python tools/span_gap.py 4096 512 > $OUT/span_gap_span.json 2> $OUT/err.log || { tail -5 $OUT/err.log; exit 1; }
python -c "
import json; d=json.load(open('$OUT/span_gap_span.json')); print(d['summary'])
for r in d['regions']: print(r['span_us'], r['span_over_fast_path_waves_us'], r['fast_path_life_us'], r['life_fit_us'])"
