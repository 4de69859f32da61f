#!/usr/bin/env python3
"""Experiment: the 4 096-env batch as TWO independent half-batch handles on two HIP streams, each fed by
its own host thread (msnake_step_tape: one launch per step, issued from C), against ONE 4 096-env handle.
What a rollout loop that double-buffers two env groups against its policy would see.  Not the bench
metric: `msnake_step` on one handle is one launch per step.   usage: two_stream_probe.py [steps]"""
import ctypes
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msnake

T = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
NS = 3


def make(n, base, stream):
    with torch.cuda.stream(stream):
        env = msnake.MultiSnakeVecEnv(n, dim=19, n_snakes=NS, seed=0, env_id_base=base, device="cuda:0")
        env.reset_device()
        tape = torch.randint(0, 5, (256, n, NS), dtype=torch.int32, device="cuda:0")
    return env, tape


def run(env, tape, steps, stream):
    L, h = env._L, env._h
    k = 0
    while k < steps:
        m = min(256, steps - k)
        msnake._capi.check(L.msnake_step_tape(h, tape.data_ptr(), NS, m, env._obs.data_ptr(), 0, env._rew.data_ptr(),
                                              env._done.data_ptr(), env._info.data_ptr(), 0, ctypes.c_void_p(stream.cuda_stream)), "step_tape")
        k += m


def timed(groups, steps):
    """groups: [(env, tape, stream)]; one host thread per group; wall time with a device sync on both sides."""
    for e, t, s in groups:
        run(e, t, 64, s)
    torch.cuda.synchronize()
    best = []
    for _ in range(7):
        th = [threading.Thread(target=run, args=(e, t, steps, s)) for e, t, s in groups]
        t0 = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        torch.cuda.synchronize()
        best.append((time.perf_counter() - t0) / steps * 1e6)
    best.sort()
    return best[len(best) // 2], best[0]


s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
one = [make(4096, 0, s0) + (s0,)]
two = [make(2048, 0, s0) + (s0,), make(2048, 2048, s1) + (s1,)]
half = [two[0]]
for name, g in (("one handle, 4096 envs, one stream", one), ("one handle, 2048 envs, one stream", half),
                ("two handles x 2048 envs, two streams", two), ("one handle, 4096 envs, one stream (again)", one)):
    med, mn = timed(g, T)
    print(f"{name:45s} {med:6.2f} us per step of the whole group set (best {mn:.2f})")
