#!/usr/bin/env python3
"""Experiment: where does the process-to-process spread of the 4 096-env step time (5.7 vs 6.1 us) come from?
One process, several observation buffers and several handles (= several physical placements of each)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msnake

n, NS, K = 4096, 3, 1024
tape = torch.randint(0, 5, (256, n, NS), dtype=torch.int32, device="cuda:0")


def us_per_step(env, obs):
    L, h = env._L, env._h
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run(m):
        k = 0
        while k < m:
            c = min(256, m - k)
            msnake._capi.check(L.msnake_step_tape(h, tape.data_ptr(), NS, c, obs.data_ptr(), 0, env._rew.data_ptr(),
                                                  env._done.data_ptr(), env._info.data_ptr(), 0, st), "step_tape")
            k += c
    run(64)
    out = []
    for _ in range(5):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(K); e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / K)
    out.sort()
    return out[2]


keep = []
envs = []
for i in range(4):
    env = msnake.MultiSnakeVecEnv(n, dim=19, n_snakes=NS, seed=0, device="cuda:0")
    env.reset_device()
    envs.append(env)
    keep.append(torch.empty(3 * 1024 * 1024 + 4096 * i, dtype=torch.uint8, device="cuda:0"))  # shift later allocations
obs_bufs = []
for i in range(4):
    obs_bufs.append(torch.empty((n, 21, 21, 9), dtype=torch.uint8, device="cuda:0"))
    keep.append(torch.empty(5 * 1024 * 1024 + 8192 * i, dtype=torch.uint8, device="cuda:0"))
print("rows: handle (its own state allocation), columns: observation buffer; us per step")
for i, env in enumerate(envs):
    print(f"handle {i}: " + "  ".join(f"{us_per_step(env, o):.2f}" for o in obs_bufs) + f"   own buffer {us_per_step(env, env._obs):.2f}")
print("obs buffer addresses: " + " ".join(hex(o.data_ptr()) for o in obs_bufs))
