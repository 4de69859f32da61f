#!/bin/bash
# Runs ON THE GPU BOX: dynamic instruction counts of kernel variants (with / without observations, render only,
# 1-3 snakes), to see where a wave's instructions go.  usage: tools/pmc_decompose.sh <tag>
set -u
TAG=${1:-dec}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for v in "step 3" "noobs 3" "render 3" "reset 3" "step 2" "step 1" "noobs 1"; do
  set -- $v
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_BRANCH SQ_INSTS_SMEM --output-format csv -d $OUT/$1_$2 -- python3 $R/tools/pmc_variant.py $1 $2 > $OUT/$1_$2.log 2>&1 || echo "$v failed"
done
python3 - <<PY
import csv, glob, collections, statistics
for v in ("step_3", "noobs_3", "render_3", "reset_3", "step_2", "step_1", "noobs_1"):
    f = glob.glob("$OUT/" + v + "/**/*counter_collection.csv", recursive=True)
    if not f:
        print(v, "no data"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        if "msnake_step_kernel" in r["Kernel_Name"]:
            m = r["Kernel_Name"].split("<")[1].split(">")[0]
            acc[m][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for m, d in acc.items():
        n = len(d["SQ_WAVES"])
        tailn = max(1, n // 2)   # the later half of the launches = steady state
        w = statistics.median(d["SQ_WAVES"][-tailn:])
        print(v, "kernel<%s>" % m, "launches", n, {k.replace("SQ_INSTS_", ""): round(statistics.median(x[-tailn:]) / w, 1) for k, x in sorted(d.items()) if k != "SQ_WAVES"})
PY
