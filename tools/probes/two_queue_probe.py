#!/usr/bin/env python3
"""Probe: does splitting one 4096-env step into two 2048-env launches on two HIP streams (two
hardware queues) shorten the wave-dispatch ramp?  Prints us per step for 1x4096 and 2x2048."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import msnake


def run(parts, iters=1280):
    n = 4096 // parts
    envs, streams, acts = [], [], []
    for i in range(parts):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            e = msnake.MultiSnakeVecEnv(n, dim=19, n_snakes=3, seed=0, env_id_base=i * n)
            e.reset_device()
            a = torch.randint(0, 5, (n, 3), dtype=torch.int32, device="cuda")
        envs.append(e); streams.append(s); acts.append(a)
    torch.cuda.synchronize()
    steps = 64
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):  # one chain of launches per stream, joined at the end
        cur = torch.cuda.current_stream()
        for e, s, a in zip(envs, streams, acts):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                for _ in range(steps):
                    e.step_device(a)
        for s in streams:
            cur.wait_stream(s)
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters // steps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / ((iters // steps) * steps) * 1e6


for parts in (1, 2, 4):
    print(f"{parts} x {4096 // parts} envs: {run(parts):.2f} us per 4096-env step")
