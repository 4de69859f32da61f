#!/usr/bin/env python3
"""SURVEY 8(d) series on one GPU: warm-up 64 steps, time 1 024 steps, 5 repeats, median.
Native observations at 4 096 / 32 768 / 262 144 envs (19x19x3), the fused x4 (84x84) series, and
BASELINE configs[1] (10x10x1) and configs[4] (19x19x2).  One msnake_step launch per step, issued
from C (msnake_step_tape); HIP events on the launch stream.  Writes one JSON document to stdout."""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import msnake

CASES = [  # name, envs, dim, snakes, obs_scale, steps
    ("19x19x3 native", 4096, 19, 3, 1, 1024),
    ("19x19x3 native", 32768, 19, 3, 1, 1024),
    ("19x19x3 native", 262144, 19, 3, 1, 256),
    ("19x19x3 fused x4 (84x84x9)", 4096, 19, 3, 4, 1024),
    ("19x19x3 fused x4 (84x84x9)", 32768, 19, 3, 4, 128),
    ("10x10x1 native (BASELINE configs[1])", 4096, 10, 1, 1, 1024),
    ("19x19x2 native (BASELINE configs[4])", 4096, 19, 2, 1, 1024),
]
out = []
dev = torch.device("cuda", 0)
for name, n, dim, ns, scale, steps in CASES:
    env = msnake.MultiSnakeVecEnv(n, dim=dim, n_snakes=ns, seed=0, device=dev, obs_scale=scale)
    env.reset_device()
    T = 64
    while T > 8 and T * n * ns * 4 > (32 << 20):  # (the action tape stays below 32 MiB: see bench.py)
        T //= 2
    rng = np.random.default_rng(1234)
    tape = torch.from_numpy(rng.integers(0, 5, size=(T, n, ns), dtype=np.int32)).to(dev)
    L, h = env._L, env._h

    def run(k):
        done_k = 0
        while done_k < k:
            m = min(T, k - done_k)
            msnake._capi.check(L.msnake_step_tape(h, tape.data_ptr(), ns, m, env._obs.data_ptr(), 0,
                                                  env._rew.data_ptr(), env._done.data_ptr(),
                                                  env._info.data_ptr(), 0, env._stream()))
            done_k += m

    run(64)
    torch.cuda.synchronize()
    us = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(steps)
        e1.record()
        torch.cuda.synchronize()
        us.append(e0.elapsed_time(e1) * 1e3 / steps)
    med = statistics.median(us)
    B = env.algorithmic_bytes_per_env_step()
    st = env.stats()
    out.append({"case": name, "envs": n, "steps_timed": steps, "repeats_us_per_step": [round(u, 2) for u in us],
                "median_us_per_step": round(med, 2), "env_steps_per_s": round(n / med * 1e6),
                "agent_steps_per_s": round(n * ns / med * 1e6), "algorithmic_bytes_per_env_step": B,
                "algorithmic_GBs": round(n * B / med / 1e3, 1), "frac_of_8TBs": round(n * B / med / 1e3 / 8000.0, 4),
                "errors": int(st["errors"])})
    del env, tape
    torch.cuda.empty_cache()
print(json.dumps({"gpu": torch.cuda.get_device_name(0), "method": __doc__.split("\n")[0], "series": out}, indent=1))
