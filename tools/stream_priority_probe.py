#!/usr/bin/env python3
"""Experiment: cadence of back-to-back msnake_step launches on the default stream, a normal side stream and
side streams of high / low priority (does the queue a stream maps to change the launch boundary?).
usage: stream_priority_probe.py [envs] [steps]"""
import ctypes
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msnake

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 256
NS = 3
env = msnake.MultiSnakeVecEnv(n, dim=19, n_snakes=NS, seed=0, device="cuda:0")
env.reset_device()
tape = torch.randint(0, 5, (K, n, NS), dtype=torch.int32, device="cuda:0")
L, h = env._L, env._h
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)


def launch(stream):
    msnake._capi.check(L.msnake_step_tape(h, tape.data_ptr(), NS, K, env._obs.data_ptr(), 0, env._rew.data_ptr(),
                                          env._done.data_ptr(), env._info.data_ptr(), 0, ctypes.c_void_p(stream.cuda_stream)), "step_tape")


def timed(stream):
    out = []
    for _ in range(9):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        launch(stream)
        e0.record(stream)
        for _ in range(4):
            launch(stream)
        e1.record(stream)
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / (4 * K))
    return round(statistics.median(out), 3)


streams = {"default": torch.cuda.default_stream(), "side": torch.cuda.Stream(), "side high priority": torch.cuda.Stream(priority=-1),
           "side low priority": torch.cuda.Stream(priority=0)}
for rnd in range(2):
    for name, st in streams.items():
        print(f"{name:20s} {timed(st)} us per step")
