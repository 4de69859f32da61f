#!/usr/bin/env python3
"""gpurun_out/prof_<tag>_rollout (tools/profile_rollout.sh) -> profiles/<tag>_rollout_tape_strided.json.
bench.py's rollout leg stores every step's observations to their own slice of a 256-step buffer
(256 x 16.3 MB), so the memory-side WRITE_SIZE of the persistent kernel is what really reaches HBM."""
import csv
import glob
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
base = f"gpurun_out/prof_{tag}_rollout"
KEY = "msnake_step_kernel<0, 3, 3"  # snake_env, 3 snakes, MODE 3 = persistent tape


def rows(pat):
    return list(csv.DictReader(open(glob.glob(base + pat, recursive=True)[0])))


import glob as _g
k = [r for r in rows("/kt/**/*_kernel_stats.csv") if KEY in r["Name"]][0]


def pmc(dirn, cname):
    v = [float(r["Counter_Value"]) for r in rows(f"/{dirn}/**/*_counter_collection.csv")
         if KEY in r["Kernel_Name"] and r["Counter_Name"] == cname]
    return sum(v) / len(v)


f, w = pmc("pmc_fetch", "FETCH_SIZE"), pmc("pmc_write", "WRITE_SIZE")
alg = 18067456
hbm = int((2 * f + w) * 1024 / 256)
us = float(k["AverageNs"]) / 256 / 1000
out = {"kernel": k["Name"], "launches": int(k["Calls"]), "steps_per_launch": 256,
       "avg_launch_ns": float(k["AverageNs"]), "us_per_step": us,
       "fetch_size_kib_per_launch": f, "write_size_kib_per_launch": w,
       "hbm_bytes_per_step": hbm, "algorithmic_bytes_per_step": alg,
       "traffic_over_algorithmic": round(hbm / alg, 4),
       "algorithmic_GBs": round(alg / us / 1e3, 1), "counter_GBs": round(hbm / us / 1e3, 1),
       "frac_of_hbm_peak_algorithmic": round(alg / us / 1e3 / 8000.0, 4),
       "note": "every step's observations go to their own slice of a 256 x 16.3 MB buffer (obs_step_stride = 16.3 MB), "
               "so each step's 16.3 MB of observation stores reaches memory; rocprofv3 --kernel-trace --stats for the "
               "duration, separate --pmc FETCH_SIZE / WRITE_SIZE passes for the traffic (FETCH doubled for gfx950)"}
json.dump(out, open(f"profiles/{tag}_rollout_tape_strided.json", "w"), indent=1)
print(json.dumps(out, indent=1))
