#!/bin/bash
# Runs ON THE GPU BOX: direct launches vs graph replay in many separate processes -- does the slow
# process mode (DESIGN "run-to-run spread") show in graph replay too?
OUT=gpurun_out/graph_modes; mkdir -p $OUT
for i in $(seq 1 ${1:-14}); do
  python tools/graph_probe.py 4096 ${2:-128} ${3:-8} 2>/dev/null | tee -a $OUT/runs.txt
done
