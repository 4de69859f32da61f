#!/bin/bash
# Runs ON THE GPU BOX: instructions executed up to each section boundary of the step kernel (diagnostic build).
set -u
TAG=${1:-stages}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp MSNAKE_LIB=$R/self-play-on-multi-snakes-environment_amd/libmsnake_dbg.so
cd /tmp
for st in 7 2 3 4 5 6 0; do
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_WAVES --output-format csv -d $OUT/st$st -- python3 $R/tools/pmc_stage.py $st > $OUT/st$st.log 2>&1 || echo "stage $st failed"
done
python3 - <<PY
import csv, glob, collections, statistics
names = {7: "prologue + loads + wait", 2: "+ moves", 3: "+ collision matrix", 4: "+ aliveness", 5: "+ stats, outputs, reset", 6: "+ paint", 0: "+ copy-out, write-back (whole kernel)"}
prev = None
for st in (7, 2, 3, 4, 5, 6, 0):
    f = glob.glob("$OUT/st%d/**/*counter_collection.csv" % st, recursive=True)
    if not f:
        print(st, "no data"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "msnake_step_kernel<0, 3, 0, 1>" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    k = 32  # the staged launches are the last 32
    w = statistics.median(acc["SQ_WAVES"][-k:])
    cur = {c.replace("SQ_INSTS_", ""): statistics.median(v[-k:]) / w for c, v in acc.items() if c != "SQ_WAVES"}
    tot = sum(cur.values())
    d = "" if prev is None else "  (section: %+.1f)" % (tot - prev)
    print("stage %d %-40s" % (st, names[st]), {k2: round(v, 1) for k2, v in sorted(cur.items())}, "total %.1f" % tot, d)
    prev = tot
PY
