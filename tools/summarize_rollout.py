#!/usr/bin/env python3
"""gpurun_out/prof_<tag>_rollout (tools/profile_rollout.sh) -> profiles/<tag>_rollout_tape.json."""
import csv
import glob
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
base = f"gpurun_out/prof_{tag}_rollout"
KEY = "msnake_step_kernel<0, 3, 3"  # snake_env, 3 snakes, MODE 3 = persistent tape


def rows(pat):
    return list(csv.DictReader(open(glob.glob(base + pat)[0])))


k = [r for r in rows("/kt/runc/*_kernel_stats.csv") if KEY in r["Name"]][0]


def pmc(dirn, cname):
    v = [float(r["Counter_Value"]) for r in rows(f"/{dirn}/runc/*_counter_collection.csv")
         if KEY in r["Kernel_Name"] and r["Counter_Name"] == cname]
    return sum(v) / len(v)


f, w = pmc("pmc_fetch", "FETCH_SIZE"), pmc("pmc_write", "WRITE_SIZE")
out = {"kernel": k["Name"], "launches": int(k["Calls"]), "steps_per_launch": 256,
       "avg_launch_ns": float(k["AverageNs"]), "us_per_step": float(k["AverageNs"]) / 256 / 1000,
       "fetch_size_kib_per_launch": f, "write_size_kib_per_launch": w,
       "hbm_bytes_per_step": int((2 * f + w) * 1024 / 256), "algorithmic_bytes_per_step": 18067456,
       "note": "bench.py re-uses one observation buffer every step (stride 0): successive steps overwrite the "
               "same 16.3 MB, the write-back L2s merge about half of those rewrites, so the memory-side "
               "WRITE_SIZE is below the 16.3 MB every step stores; FETCH is the 48 KB action row per step"}
json.dump(out, open(f"profiles/{tag}_rollout_tape.json", "w"), indent=1)
print(json.dumps(out, indent=1))
