#!/usr/bin/env python3
"""How long does the HOST take to enqueue one msnake_step launch (msnake_step_tape, C loop), next to the GPU time
per step of the same launches?  If the two are close, a slow host thread makes the measurement host-bound."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msnake

n, NS, K = 4096, 3, 1024
env = msnake.MultiSnakeVecEnv(n, dim=19, n_snakes=NS, seed=0, device="cuda:0")
env.reset_device()
tape = torch.randint(0, 5, (256, n, NS), dtype=torch.int32, device="cuda:0")
L, h = env._L, env._h
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(m):
    k = 0
    while k < m:
        c = min(256, m - k)
        msnake._capi.check(L.msnake_step_tape(h, tape.data_ptr(), NS, c, env._obs.data_ptr(), 0, env._rew.data_ptr(),
                                              env._done.data_ptr(), env._info.data_ptr(), 0, st), "step_tape")
        k += c


run(64)
torch.cuda.synchronize()
rows = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record(); run(K); e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    rows.append(((t1 - t0) / K * 1e6, e0.elapsed_time(e1) * 1e3 / K))
print("host enqueue us per launch / GPU us per step: " + "  ".join(f"{a:.2f}/{b:.2f}" for a, b in rows), "| cpu", os.sched_getaffinity(0).__len__(), "cores")
