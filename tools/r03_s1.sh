#!/bin/bash
# Runs ON THE GPU BOX (round 3, session 1): parity suite, bench with its new legs, same-box A/B of the late Philox
# refill, the span / gap decomposition, the traffic split, the process-mode samples, a 1-rank RCCL bench.
set -u
TAG=${1:-r03a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
PKG=$R/self-play-on-multi-snakes-environment_amd
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
guard() {  # a GPU step that was killed at its limit ends the session (no further GPU step after a hang)
  "$@"; local rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL in: $*"; exit 1; fi
  return $rc
}
guard timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?
tail -6 $OUT/pytest.log
[ $rc -ne 0 ] && { grep -E "Error|assert|FAILED" $OUT/pytest.log | head -20; exit $rc; }
echo "== bench"
guard timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $OUT/bench_k20.json 2> $OUT/bench_k20.err || { tail -5 $OUT/bench_k20.err; exit 1; }
guard timeout -k 10 400 python bench.py --steps 1024 --warmup 64 --no-cpu-baseline > $OUT/bench_k1024.json 2> $OUT/bench_k1024.err || { tail -5 $OUT/bench_k1024.err; exit 1; }
python - <<PY
import json
for f in ("bench_k20", "bench_k1024"):
    d = json.load(open("$OUT/" + f + ".json"))
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], "tape", d.get("rollout_tape", {}).get("us_per_step"),
          "strided", d.get("per_step_strided", {}).get("us_per_step"), "selfplay", d.get("selfplay_rollout"), "cpu", d.get("cpu_baseline", {}).get("value"), d.get("collective"))
PY
echo "== A/B late Philox refill (us per launch, 4096 envs)"
for i in 1 2 3 4; do
  for lib in "" $PKG/libmsnake_nolate.so; do
    MSNAKE_LIB=$lib guard timeout -k 10 200 python bench.py --steps 1024 --warmup 64 --repeats 5 --no-cpu-baseline --no-rollout 2>/dev/null > $OUT/ab.json
    python -c "
import json; d=json.load(open('$OUT/ab.json')); print('${lib:-late(default)}'.split('/')[-1], d['roofline']['launch_us'], d['timing']['repeats_us_per_step'])" | tee -a $OUT/ab_late_refill.txt
  done
done
echo "== span / gap"
MSNAKE_LIB=$PKG/libmsnake_dbg.so guard timeout -k 10 300 python tools/span_gap.py 4096 512 > $OUT/span_gap_4096.json 2> $OUT/span_gap.err || tail -5 $OUT/span_gap.err
python -c "
import json; d=json.load(open('$OUT/span_gap_4096.json')); print(d['summary']); print(d['regions'][2])"
# the diagnostic build's own HIP-event figure without the span stamps, and the production library, same box
MSNAKE_LIB=$PKG/libmsnake_dbg.so guard timeout -k 10 200 python bench.py --steps 1024 --warmup 64 --repeats 5 --no-cpu-baseline --no-rollout 2>/dev/null > $OUT/bench_dbg_nostamps.json
guard timeout -k 10 200 python bench.py --steps 1024 --warmup 64 --repeats 5 --no-cpu-baseline --no-rollout 2>/dev/null > $OUT/bench_prod_same_box.json
( cd /tmp && guard timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py --steps 1024 --warmup 64 --repeats 1 --no-cpu-baseline --no-rollout > $OUT/kt.log 2>&1 )
python - <<PY
import json, glob, csv
for f in ("bench_dbg_nostamps", "bench_prod_same_box"):
    print(f, json.load(open("$OUT/" + f + ".json"))["roofline"]["launch_us"])
for f in glob.glob("$OUT/kt/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:3]:
        print(r["Name"][:50], r["Calls"], r["AverageNs"])
PY
echo "== traffic split"
guard bash tools/pmc_traffic_split.sh $TAG/split262144 262144 > $OUT/split_262144.json 2>&1; tail -30 $OUT/split_262144.json
guard bash tools/pmc_traffic_split.sh $TAG/split4096 4096 > $OUT/split_4096.json 2>&1; tail -30 $OUT/split_4096.json
echo "== process modes"
guard bash tools/proc_mode.sh 12 > $OUT/proc_mode.jsonl 2>&1
python - <<PY
import json
for l in open("$OUT/proc_mode.jsonl"):
    if l.startswith("{"):
        d = json.loads(l); print(d["us_per_step"], d["first_launch_us"], d["tiny_kernel_us"], d["stream_handle"], d["obs_ptr"], d["smi_before"].get("sclk_mhz"), d["smi_during"].get("sclk_mhz"), d["smi_during"].get("power_w"))
PY
echo "== 1-rank RCCL bench under torchrun"
guard timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --backend nccl --steps 1024 --warmup 64 --no-cpu-baseline > $OUT/bench_1rank_rccl.json 2> $OUT/bench_1rank_rccl.err || tail -5 $OUT/bench_1rank_rccl.err
grep -o '"collective": {[^}]*}' $OUT/bench_1rank_rccl.json
echo "session done"
