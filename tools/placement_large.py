#!/usr/bin/env python3
"""Does the 262 144-env launch time depend on WHERE its 1 GB observation tensor lies?  One process, one handle, several
observation buffers (separated by allocations of odd sizes), the same steps timed into each; then the same for a second
handle.  usage: python tools/placement_large.py [envs]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import msnake

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
NS, T = 3, 32
tape = torch.from_numpy(np.random.default_rng(1234).integers(0, 5, (T, n, NS)).astype(np.int32)).cuda()


def us_per_step(env, obs, iters=96):
    L, h = env._L, env._h
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run(m):
        k = 0
        while k < m:
            c = min(T, m - k)
            msnake._capi.check(L.msnake_step_tape(h, tape.data_ptr(), NS, c, obs.data_ptr(), 0, env._rew.data_ptr(),
                                                  env._done.data_ptr(), env._info.data_ptr(), 0, st), "step_tape")
            k += c
    run(16)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(iters); e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


keep, bufs = [], []
for i in range(5):
    bufs.append(torch.empty((n, 21, 21, 9), dtype=torch.uint8, device="cuda"))
    keep.append(torch.empty(37 * 1024 * 1024 + 4096 * (i + 1), dtype=torch.uint8, device="cuda"))
for hidx in range(2):
    env = msnake.MultiSnakeVecEnv(n, dim=19, n_snakes=NS, seed=hidx)
    env.reset_device()
    print(f"handle {hidx}: own buffer {us_per_step(env, env._obs):.1f} us; " + "  ".join(f"{us_per_step(env, b):.1f}" for b in bufs), flush=True)
    keep.append(env)
print("buffer addresses:", " ".join(hex(b.data_ptr()) for b in bufs))
