#!/bin/bash
# Runs ON THE GPU BOX: dynamic instruction mix of the step kernel (one rocprofv3 --pmc pass of the bench
# workload) + the plain bench line, for A/B runs of kernel variants.  usage: tools/pmc_insts.sh <tag> [envs]
set -u
TAG=${1:-ab}
ENVS=${2:-4096}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 512 --warmup 64 --repeats 3 --envs-per-gpu $ENVS --no-cpu-baseline"
python3 $R/bench.py $ARGS > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d $OUT/pmc_insts -- python3 $R/bench.py $ARGS --no-rollout > $OUT/pmc.log 2>&1 || echo "pmc pass failed"
python3 - <<PY
import csv, glob, json, collections, statistics
d = json.load(open("$OUT/bench.json"))
print("bench: us/step", d["roofline"]["launch_us"], "frac", d["roofline"]["frac"], "rollout us/step", d.get("rollout_tape", {}).get("us_per_step"))
f = glob.glob("$OUT/pmc_insts/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if ", 0, 1>" in r["Kernel_Name"] and "msnake_step_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
w = statistics.median(acc["SQ_WAVES"])
print("per wave:", {k.replace("SQ_INSTS_", ""): round(statistics.median(v) / w, 1) for k, v in sorted(acc.items())})
PY
