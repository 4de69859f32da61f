#!/usr/bin/env python3
"""Fused x4 (84x84x9) step time by batch size and store policy (HIP events, msnake_step_tape, one box).
usage: python tools/fused_ab.py [envs ...]"""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import msnake

sizes = [int(a) for a in sys.argv[1:]] or [4096, 32768]
for n in sizes:
    for pol in ("plain", "stream", "auto"):
        env = msnake.MultiSnakeVecEnv(n, dim=19, n_snakes=3, seed=0, obs_scale=4, obs_store_policy=pol)
        env.reset_device()
        T = 32
        while T > 8 and T * n * 3 * 4 > (32 << 20):
            T //= 2
        tape = torch.from_numpy(np.random.default_rng(1234).integers(0, 5, (T, n, 3)).astype(np.int32)).cuda()
        L, h = env._L, env._h
        steps = 512 if n <= 8192 else 96

        def run(k):
            done = 0
            while done < k:
                m = min(T, k - done)
                msnake._capi.check(L.msnake_step_tape(h, tape.data_ptr(), 3, m, env._obs.data_ptr(), 0, env._rew.data_ptr(),
                                                      env._done.data_ptr(), env._info.data_ptr(), 0, env._stream()))
                done += m
        run(32)
        torch.cuda.synchronize()
        us = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(steps); e1.record()
            torch.cuda.synchronize()
            us.append(e0.elapsed_time(e1) * 1e3 / steps)
        med = statistics.median(us)
        B = env.algorithmic_bytes_per_env_step() * n
        print(json.dumps({"envs": n, "policy": pol, "us_per_step": round(med, 2), "frac_of_8TBs": round(B / med / 1e3 / 8000, 4),
                          "obs_MiB": round(n * 84 * 84 * 9 / 2**20)}), flush=True)
        env.close()
        del env, tape
        torch.cuda.empty_cache()
