#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
PKG=$R/self-play-on-multi-snakes-environment_amd
cd $R
mkdir -p gpurun_out/r03m
export TMPDIR=/tmp
show() { grep '^{' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  envs', d['envs'], 'step', d['step_us'], 'alg_GBs', d['alg_GBs'], 'frac', round(d['alg_GBs']/8000, 3), 'render', d['render_us'], 'reset', d['reset_us'])"; }
MSNAKE_LIB=$PKG/libmsnake_fullnt.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ragged or large_batches_byte or config3 or golden_tape" 2>&1 | tail -2
for rep in 1 2; do
for lib in default fullnt; do
  L=$PKG/libmsnake_$lib.so; [ $lib = default ] && L=""
  echo "lib=$lib"
  MSNAKE_LIB=$L timeout -k 10 300 python tools/kbench.py --envs 262144 --iters 120 2>/dev/null | show
  MSNAKE_LIB=$L timeout -k 10 300 python tools/kbench.py --envs 32768 65536 131072 262144 --iters 120 2>/dev/null | show
done
done
echo "bench.py at 262144 / 65536 (fullnt, default)"
for lib in fullnt default; do
  L=$PKG/libmsnake_$lib.so; [ $lib = default ] && L=""
  for n in 262144 65536; do
  MSNAKE_LIB=$L timeout -k 10 300 python bench.py --envs-per-gpu $n --steps 64 --warmup 16 --repeats 5 --no-cpu-baseline --no-rollout 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('$lib', $n, d['roofline']['launch_us'], d['roofline']['frac'], d['timing']['repeats_us_per_step'])"
  done
done
MSNAKE_LIB=$PKG/libmsnake_fullnt.so bash tools/pmc_traffic_split.sh r03m/split_fullnt 262144 2>&1 | grep -A3 '"step <0, 3, 0, 1>"\|"render <0, 3, 2, 1>"'
