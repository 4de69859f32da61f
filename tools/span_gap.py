#!/usr/bin/env python3
"""Diagnostic (MSNAKE_DBG_STAGES build only): what a per-step launch cadence is made of, on the GPU's own clock.

K back-to-back msnake_step launches (one stream, issued from C); every wave stamps s_memrealtime (100 MHz, one
clock for the whole chip) into the launch's own slot when it starts, when it has issued its last store and when all
its stores have been acknowledged.  Per launch: span = max(acknowledged) - min(start); gap to the next launch =
its min(start) - this max(acknowledged); span + gap = the cadence, which the same run also measures with HIP
events around the K launches.  Prints one JSON document.

usage: MSNAKE_LIB=<pkg>/libmsnake_dbg.so python tools/span_gap.py [envs] [K]"""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 512
span = torch.zeros((K, n, 4), dtype=torch.int64, device="cuda")
os.environ["MSNAKE_DBG_SPAN"] = hex(span.data_ptr())
os.environ["MSNAKE_DBG_SPAN_SLOTS"] = str(K)
import msnake

assert "dbg" in os.path.basename(msnake._capi.LIB_PATH), "needs the diagnostic build: MSNAKE_LIB=.../libmsnake_dbg.so"
env = msnake.MultiSnakeVecEnv(n, dim=19, n_snakes=3, seed=0)
env.reset_device()
T = 256
tape = torch.from_numpy(np.random.default_rng(1234).integers(0, 5, (T, n, 3)).astype(np.int32)).cuda()
L, h = env._L, env._h


def run(k, start):
    done = 0
    while done < k:
        off = (start + done) % T
        m = min(k - done, T - off)
        msnake._capi.check(L.msnake_step_tape(h, tape[off].data_ptr(), 3, m, env._obs.data_ptr(), 0, env._rew.data_ptr(),
                                              env._done.data_ptr(), env._info.data_ptr(), 0, env._stream()))
        done += m


run(K, 0)  # warm-up: K launches (slots 0..K-1), overwritten below
torch.cuda.synchronize()
reps = []
for rep in range(5):
    span.zero_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    env.render_device()  # (like bench.py: the first event fires behind a running kernel)
    e0.record()
    run(K, K * (rep + 1))  # the launch counter is at a multiple of K: launch i of this region stamps slot i
    e1.record()
    torch.cuda.synchronize()
    ev_us = e0.elapsed_time(e1) * 1e3 / K
    s = span.cpu().numpy()
    start, issued, acked = s[:, :, 0], s[:, :, 1], s[:, :, 2]
    assert (start > 0).all() and (acked >= issued).all()
    first, last_start = start.min(1), start.max(1)
    end_issue, end_ack = issued.max(1), acked.max(1)
    spans = (end_ack - first) / 100.0
    gaps = (first[1:] - end_ack[:-1]) / 100.0
    cad = (first[1:] - first[:-1]) / 100.0
    reps.append({
        "hip_event_us_per_launch": round(ev_us, 3),
        "stamp_cadence_us": round(float((first[-1] - first[0]) / 100.0 / (K - 1)), 3),
        "span_us": {"median": round(float(np.median(spans)), 3), "p10": round(float(np.percentile(spans, 10)), 3),
                    "p90": round(float(np.percentile(spans, 90)), 3)},
        "gap_us": {"median": round(float(np.median(gaps)), 3), "p10": round(float(np.percentile(gaps, 10)), 3),
                   "p90": round(float(np.percentile(gaps, 90)), 3), "negative_share": round(float((gaps < 0).mean()), 4)},
        "cadence_us_median": round(float(np.median(cad)), 3),
        "first_to_last_wave_start_us": round(float(np.median((last_start - first) / 100.0)), 3),
        "first_start_to_last_store_issued_us": round(float(np.median((end_issue - first) / 100.0)), 3),
        "last_store_issued_to_last_ack_us": round(float(np.median((end_ack - end_issue) / 100.0)), 3),
        "wave_life_us_median": round(float(np.median((acked - start) / 100.0)), 3),
    })
med = lambda k: statistics.median(r[k] for r in reps)
out = {"what": __doc__.split("\n")[0], "gpu": torch.cuda.get_device_name(0), "library": os.path.basename(msnake._capi.LIB_PATH),
       "kernel": env.kernel_name(), "envs": n, "launches_per_region": K, "clock": "s_memrealtime, 100 MHz, 10 ns per tick",
       "regions": reps,
       "summary": {"hip_event_us_per_launch": med("hip_event_us_per_launch"), "stamp_cadence_us": med("stamp_cadence_us"),
                   "span_us": statistics.median(r["span_us"]["median"] for r in reps),
                   "gap_us": statistics.median(r["gap_us"]["median"] for r in reps)}}
out["summary"]["span_plus_gap_us"] = round(out["summary"]["span_us"] + out["summary"]["gap_us"], 3)
out["summary"]["span_plus_gap_over_hip_event"] = round(out["summary"]["span_plus_gap_us"] / out["summary"]["hip_event_us_per_launch"], 4)
print(json.dumps(out, indent=1))
