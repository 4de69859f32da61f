#!/bin/bash
# Runs ON THE GPU BOX: N fresh processes of tools/proc_mode_probe.py, one JSON line each.
N=${1:-12}
for i in $(seq 1 $N); do
  timeout -k 10 120 python tools/proc_mode_probe.py 2>/dev/null | grep '^{"us_per_step"'
done
