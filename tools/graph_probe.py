#!/usr/bin/env python3
"""Experiment: K per-step launches of msnake_step replayed from ONE captured HIP graph against the same
launches issued directly (msnake_step_tape, C loop).   usage: graph_probe.py [envs] [steps per graph] [replays per timed region]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msnake

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 128
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 8
NS = 3
env = msnake.MultiSnakeVecEnv(n, dim=19, n_snakes=NS, seed=0, device="cuda:0")
env.reset_device()
tape = torch.randint(0, 5, (K, n, NS), dtype=torch.int32, device="cuda:0")
L, h = env._L, env._h


def launch(stream):
    msnake._capi.check(L.msnake_step_tape(h, tape.data_ptr(), NS, K, env._obs.data_ptr(), 0, env._rew.data_ptr(),
                                          env._done.data_ptr(), env._info.data_ptr(), 0, ctypes.c_void_p(stream.cuda_stream)), "step_tape")


side = torch.cuda.Stream()
with torch.cuda.stream(side):
    launch(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    launch(side)
torch.cuda.synchronize()


def timed(fn, reps=REPS):
    out = []
    for _ in range(9):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(side):
            e0.record(side)
            for _ in range(reps):
                fn()
            e1.record(side)
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / (reps * K))
    out.sort()
    return out[len(out) // 2]


def direct():
    launch(side)


def replay():
    with torch.cuda.stream(side):
        g.replay()


print(f"{n} envs, {K} steps per graph, {REPS} per timed region: direct launches {timed(direct):.2f} us per step, graph replay {timed(replay):.2f} us per step")
