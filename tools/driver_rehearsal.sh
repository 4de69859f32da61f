#!/bin/bash
# Runs ON THE GPU BOX: what the driver runs at round end, with wall times (smoke, GPU tests, bench with its defaults
# and with --steps 20 --warmup 5).
OUT=gpurun_out/rehearsal; mkdir -p $OUT
t0=$SECONDS
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1 || { tail -5 $OUT/smoke.log; exit 1; }
echo "smoke $((SECONDS-t0)) s"; t0=$SECONDS
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/gpu_tests.log 2>&1 || { tail -20 $OUT/gpu_tests.log; exit 1; }
tail -1 $OUT/gpu_tests.log; echo "gpu tests $((SECONDS-t0)) s"; t0=$SECONDS
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err || { tail -5 $OUT/bench_steps20.err; exit 1; }
echo "bench --steps 20: $((SECONDS-t0)) s"; t0=$SECONDS
timeout -k 10 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -5 $OUT/bench_default.err; exit 1; }
echo "bench defaults: $((SECONDS-t0)) s"
python - <<'PY'
import json
for f in ("bench_steps20", "bench_default"):
    d = json.loads([l for l in open(f"gpurun_out/rehearsal/{f}.json") if l.startswith('{"metric"')][-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("selfplay_rollout", {}).get("frames_per_s"))
PY
