#!/bin/bash
# Runs ON THE GPU BOX: the reference's two published learning curves on the round-3 kernel (statistical parity, f3)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_curves
mkdir -p $OUT
cd $R
timeout -k 10 500 python tools/train_selfplay.py --envs 256 --snakes 2 --timesteps 10000000 --csv $OUT/ppo_selfplay_256envs_1e7steps.csv | awk 'NR % 40 == 1' | cut -c1-200
timeout -k 10 500 python tools/train_selfplay.py --envs 256 --snakes 3 --dim 10 --timesteps 20000000 --csv $OUT/ppo_selfplay_3snakes_10x10_2e7steps.csv | awk 'NR % 80 == 1' | cut -c1-200
timeout -k 10 300 python tools/train_selfplay.py --envs 4096 --snakes 2 --timesteps 5300000 --csv $OUT/ppo_selfplay_4096envs_fp32_20updates.csv | cut -c1-200
