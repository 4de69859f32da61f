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
span = torch.zeros((K, n, 8), dtype=torch.int64, device="cuda")
os.environ["MSNAKE_DBG_SPAN"] = hex(span.data_ptr())
os.environ["MSNAKE_DBG_SPAN_SLOTS"] = str(K)
import msnake

assert any(k in os.path.basename(msnake._capi.LIB_PATH) for k in ("dbg", "span")), "needs a measurement build: MSNAKE_LIB=.../libmsnake_dbg.so or libmsnake_span.so"
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
    start, loads, logic, painted, issued, acked, flags = (s[:, :, i] for i in range(7))
    light = bool((issued == 0).all())  # -DMSNAKE_SPAN_LIGHT: the production kernel with the start / acknowledged stamps only
    if light:
        loads = logic = painted = start
        issued = acked
    assert (start > 0).all() and (acked >= issued).all()
    first, last_start = start.min(1), start.max(1)
    end_issue, end_ack = issued.max(1), acked.max(1)
    spans = (end_ack - first) / 100.0
    gaps = (first[1:] - end_ack[:-1]) / 100.0
    cad = (first[1:] - first[:-1]) / 100.0
    life = (acked - start) / 100.0
    rel_end = (acked - first[:, None]) / 100.0          # when a wave ended, since its launch's first wave started
    stage = lambda a, b: round(float(np.median((b - a) / 100.0)), 3)
    pct = lambda x, q: round(float(np.percentile(x, q)), 3)
    # the waves that end a launch: its last 32 finishers
    order = np.argsort(acked, axis=1)[:, -32:]
    take = lambda a: np.take_along_axis(a, order, axis=1)
    tf = take(flags)
    tail = {"start_offset_us": round(float(np.median((take(start) - first[:, None]) / 100.0)), 3),
            "life_us": round(float(np.median(take(life))), 3),
            "loads_us": stage(take(start), take(loads)), "logic_us": stage(take(loads), take(logic)),
            "paint_us": stage(take(logic), take(painted)), "copy_out_us": stage(take(painted), take(issued)),
            "ack_us": stage(take(issued), take(acked)),
            "share_ate": round(float(((tf & 1) != 0).mean()), 3), "share_episode_end": round(float(((tf & 2) != 0).mean()), 3),
            "share_philox_ahead_of_logic": round(float((((tf & 4) != 0) & ((tf & 8) == 0)).mean()), 3),
            "share_late_refill": round(float(((tf & 8) != 0).mean()), 3),
            "share_fast_path": round(float(((tf & 0xF) == 0).mean()), 3)}
    xcc = (flags >> 8) & 15
    per_xcd = [round(float(np.median(np.where(xcc == x, rel_end, 0).max(1))), 3) for x in range(8)]
    cls = lambda m: {"share": round(float(m.mean()), 4), "life_us_median": round(float(np.median(life[m])), 3) if m.any() else None,
                     "life_us_p90": pct(life[m], 90) if m.any() else None}
    reps.append({
        "stamps": "start + acknowledged only (stage splits are not measured)" if light else "all",
        "hip_event_us_per_launch": round(ev_us, 3),
        "stamp_cadence_us": round(float((first[-1] - first[0]) / 100.0 / (K - 1)), 3),
        "span_us": {"median": round(float(np.median(spans)), 3), "p10": pct(spans, 10), "p90": pct(spans, 90)},
        "gap_us": {"median": round(float(np.median(gaps)), 3), "p10": pct(gaps, 10), "p90": pct(gaps, 90),
                   "negative_share": round(float((gaps < 0).mean()), 4)},
        "cadence_us_median": round(float(np.median(cad)), 3),
        "first_to_last_wave_start_us": round(float(np.median((last_start - first) / 100.0)), 3),
        "first_start_to_last_store_issued_us": round(float(np.median((end_issue - first) / 100.0)), 3),
        "last_store_issued_to_last_ack_us": round(float(np.median((end_ack - end_issue) / 100.0)), 3),
        "wave_life_us": {"p50": pct(life, 50), "p90": pct(life, 90), "p99": pct(life, 99), "max_median_over_launches": round(float(np.median(life.max(1))), 3)},
        "wave_stages_us_median": {"loads": stage(start, loads), "logic": stage(loads, logic), "paint": stage(logic, painted),
                                  "copy_out": stage(painted, issued), "ack": stage(issued, acked)},
        "by_class": {"fast_path": cls((flags & 0xF) == 0), "ate": cls((flags & 1) != 0), "episode_end": cls((flags & 2) != 0),
                     "philox_ahead_of_logic": cls(((flags & 4) != 0) & ((flags & 8) == 0)), "late_refill": cls((flags & 8) != 0)},
        # what the slow paths cost the launch: its span counted over the fast-path waves only
        "span_over_fast_path_waves_us": {"median": round(float(np.median(np.where((flags & 0xF) == 0, rel_end, 0).max(1))), 3),
                                         "vs_all_waves_median": round(float(np.median(rel_end.max(1) - np.where((flags & 0xF) == 0, rel_end, 0).max(1))), 3)},
        "fast_path_life_us": {"p99": pct(life[(flags & 0xF) == 0], 99), "p999": pct(life[(flags & 0xF) == 0], 99.9)},
        # least squares over all waves of the region: life = base + a*[ate] + b*[episode end] + c*[Philox ahead of the logic]
        "life_fit_us": dict(zip(("base", "ate", "episode_end", "philox"), [round(float(x), 3) for x in np.linalg.lstsq(
            np.stack([np.ones(life.size), ((flags & 1) != 0).ravel(), ((flags & 2) != 0).ravel(), (((flags & 4) != 0) & ((flags & 8) == 0)).ravel()], 1).astype(np.float64),
            life.ravel().astype(np.float64), rcond=None)[0]])),
        "last_32_finishers_per_launch": tail,
        "last_ack_per_xcd_us_since_launch_start": per_xcd,
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
