#!/usr/bin/env python3
"""profiles/<tag>_span_gap_4096.json from one tools/final_round.sh run (gpurun_out/<tag>/span_gap_{dbg,light}.json, its bench
lines and the rocprofv3 kernel stats of the same box).   usage: compose_span_gap.py <tag>"""
import csv
import json
import os
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out", tag)


def bench_line(path):
    return json.loads([l for l in open(path) if l.startswith('{"metric"')][-1])


def build(path, library):
    d = json.load(open(path))
    s = d["summary"]
    return {"library": library, **s, "region_detail": d["regions"][len(d["regions"]) // 2]}


dbg = build(os.path.join(out, "span_gap_dbg.json"), "libmsnake_dbg.so (-DMSNAKE_DBG_STAGES)")
light = build(os.path.join(out, "span_gap_light.json"), "libmsnake_span.so (-DMSNAKE_SPAN_LIGHT)")
k1024 = bench_line(os.path.join(out, "bench_k1024.json"))["roofline"]["launch_us"]
k20 = bench_line(os.path.join(out, "bench_k20.json"))["roofline"]["launch_us"]
prof = bench_line(os.path.join(root, "gpurun_out", f"prof_{tag}_4096", "bench_plain.json"))["roofline"]["launch_us"]
rp = None
for r in csv.DictReader(open(os.path.join(root, "profiles", f"{tag}_4096_kernel_stats.csv"))):
    if "msnake_step_kernel<0, 3, 0, 1>" in r["Name"]:
        rp = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
gap = round((dbg["gap_us"] + light["gap_us"]) / 2, 3)
alg = 4411 * 4096
rec = {
    "what": "What the cadence of back-to-back msnake_step launches (4 096 envs x 19x19 x 3 snakes) is made of, on the GPU's own 100 MHz "
            f"clock; everything below comes from ONE box and one command (tools/final_round.sh {tag})",
    "method": "every wave stamps s_memrealtime into its launch's own slot: start, (stages), all stores acknowledged; per launch span = "
              "max(acknowledged) - min(start), gap = next launch's min(start) - this launch's max(acknowledged); 5 regions of 512 launches "
              "each, medians (tools/span_gap.py)",
    "diagnostic_build_all_stamps": dbg,
    "production_kernel_plus_start_and_ack_stamps": light,
    "production_library_same_box": {"bench_steps1024_us_per_launch": k1024, "bench_steps20_us_per_launch": k20,
                                    "profile_process_us_per_launch": prof,
                                    "rocprofv3_kernel_trace_average_us": round(rp[0], 3) if rp else None, "rocprofv3_calls": rp[1] if rp else None},
    "reading": {
        "gap_us": gap,
        "gap_note": "the inter-kernel gap is the same in both stamped builds although their kernel spans differ: it belongs to the launch "
                    "boundary (end-of-kernel write-back and release, barrier, next dispatch), not to the kernel",
        "production_span_us_estimate": round(k1024 - gap, 3),
        "production_span_note": "HIP-event cadence of the unstamped production library minus that gap; the stamps themselves cost 0.9 (two) "
                                "to 1.5 us (all) per launch, so the stamped spans are upper bounds",
        "roofline_frac_by_cadence": round(alg / (k1024 * 1e-6) / 8e12, 4),
        "roofline_frac_by_rocprofv3_duration": round(alg / (rp[0] * 1e-6) / 8e12, 4) if rp else None,
        "roofline_frac_of_kernel_span_estimate": round(alg / ((k1024 - gap) * 1e-6) / 8e12, 4),
    },
}
path = os.path.join(root, "profiles", f"{tag}_span_gap_4096.json")
json.dump(rec, open(path, "w"), indent=1)
print(path, rec["reading"], rec["production_library_same_box"], {k: dbg[k] for k in ("span_us", "gap_us", "hip_event_us_per_launch")},
      {k: light[k] for k in ("span_us", "gap_us", "hip_event_us_per_launch")})
