#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the GPU parity suite, then the bench at the driver's step count
# and at 1 024 steps (same box), outputs under gpurun_out/<tag>/.   usage: tools/gpu_check.sh <tag> [pytest -k expr]
set -u
TAG=${1:-check}
K=${2:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
if [ -n "$K" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "$K" > $OUT/pytest.log 2>&1
else
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
fi
rc=$?
tail -15 $OUT/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_k20.json 2> $OUT/bench_k20.err || { tail -5 $OUT/bench_k20.err; exit 1; }
timeout -k 10 300 python bench.py --steps 1024 --warmup 64 --no-cpu-baseline > $OUT/bench_k1024.json 2> $OUT/bench_k1024.err || { tail -5 $OUT/bench_k1024.err; exit 1; }
python - <<PY
import json
for f in ("bench_k20", "bench_k1024"):
    d = json.load(open("$OUT/" + f + ".json"))
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["timing"]["wall_ms_per_step"], d.get("rollout_tape", {}).get("us_per_step"), d.get("cpu_baseline", {}).get("value"))
PY
