#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
PKG=$R/self-play-on-multi-snakes-environment_amd
cd $R
echo "== span-light (start stamp held in registers)"
bash tools/proc_mode_span.sh 8 | tee gpurun_out/proc_mode_span.jsonl | cut -c1-160
echo "== A/B: default vs endwait, 10 rounds"
bash tools/ab_libs.sh 10 "" default $PKG/libmsnake_endwait.so
