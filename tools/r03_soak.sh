#!/bin/bash
# Runs ON THE GPU BOX: long bit-exact runs of the round-3 kernel against the oracle (both copy-out shapes, all rule sets,
# a large batch, and play under a trained policy: long bodies, many eating steps)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
{
timeout -k 10 600 python tools/soak.py --envs 4096 --steps 2500 --rules snake_env || exit 1
timeout -k 10 600 python tools/soak.py --envs 4096 --steps 2000 --rules snake_env --store-policy plain || exit 1
timeout -k 10 600 python tools/soak.py --envs 16384 --steps 600 --rules snake_env || exit 1
timeout -k 10 600 python tools/soak.py --envs 4096 --steps 2000 --rules adversarial --dim 10 --store-policy plain || exit 1
timeout -k 10 600 python tools/soak.py --envs 4096 --steps 2000 --rules new_world --dim 10 --snakes 4 --fruits 6 --store-policy plain || exit 1
timeout -k 10 600 python tools/soak.py --envs 4096 --steps 1500 --rules adversarial || exit 1
timeout -k 10 600 python tools/soak.py --envs 65536 --steps 200 --rules snake_env --obs-every 20 || exit 1
timeout -k 10 600 python tools/soak.py --envs 4096 --steps 2048 --rules snake_env --tape 64 || exit 1
timeout -k 10 600 python tools/soak.py --envs 1000 --steps 2048 --rules snake_env --tape 64 || exit 1
timeout -k 10 600 python tools/soak.py --envs 4096 --steps 1024 --rules adversarial --dim 10 --tape 32 || exit 1
timeout -k 10 900 python tools/soak_policy.py --envs 256 --steps 4000 || exit 1
} 2>&1 | grep -v "amdgpu.ids\|MIOpen" | tee gpurun_out/r03_soak.txt
