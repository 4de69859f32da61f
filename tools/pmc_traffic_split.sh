#!/bin/bash
# Runs ON THE GPU BOX: where do the memory-side bytes of a launch go?  FETCH_SIZE / WRITE_SIZE (separate passes) of the
# step kernel with observations, without them (obs_dev = NULL: record, body-ring sectors, reward / done / info only) and
# of the render-only kernel (the observation stores alone).  usage: tools/pmc_traffic_split.sh <tag> [envs]
set -u
TAG=${1:-split}
ENVS=${2:-262144}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for v in step noobs render; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $OUT/${v}_$c -- python3 $R/tools/pmc_variant.py $v 3 $ENVS > $OUT/${v}_$c.log 2>&1 || echo "$v $c failed"
  done
done
python3 - <<PY
import csv, glob, collections, statistics, json
res = {}
for v in ("step", "noobs", "render"):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob("$OUT/%s_%s/**/*counter_collection.csv" % (v, c), recursive=True)
        if not f:
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f[0])):
            if "msnake_step_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                acc[r["Kernel_Name"].split("<")[1].split(">")[0]].append(float(r["Counter_Value"]))
        for m, x in acc.items():
            x = x[len(x) // 2:]  # steady state
            kib = statistics.fmean(x)
            b = kib * 1024 * (2 if c == "FETCH_SIZE" else 1)  # gfx950: FETCH_SIZE tallies 128-B requests at 64 B
            res.setdefault("%s <%s>" % (v, m), {})[c + "_bytes_per_env"] = round(b / $ENVS, 1)
print(json.dumps({"envs": $ENVS, "per_env": res}, indent=1))
PY
