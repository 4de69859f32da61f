#!/bin/bash
# needs a diagnostic build (make -C <pkg>/csrc -B EXTRA=-DMSNAKE_DBG_STAGES)
# timing-only: cumulative kernel time up to each debug stage (MSNAKE_DBG_STAGE), 4096 envs
for st in 1 2 3 4 0; do
  echo "stage $st: $(MSNAKE_DBG_STAGE=$st python tools/kbench.py --envs 4096 --iters 200 2>/dev/null | tail -1)"
done
