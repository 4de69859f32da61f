#!/usr/bin/env python3
"""Follow-up of tools/placement_large.py: does the ORDER of allocations in a fresh process decide how fast the 262 144-env
launch runs?  usage: python tools/placement_order.py <first|after_keep|after_free>"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import msnake

mode = sys.argv[1] if len(sys.argv) > 1 else "first"
n, NS, T = 262144, 3, int(os.environ.get("TAPE_T", "32"))
torch.cuda.init()
dummy = None
if mode in ("after_keep", "after_free"):
    dummy = [torch.empty(1 << 30, dtype=torch.uint8, device="cuda") for _ in range(3)]
    for d in dummy:
        d.fill_(1)
    torch.cuda.synchronize()
    if mode == "after_free":
        dummy = None
        torch.cuda.empty_cache()
env = msnake.MultiSnakeVecEnv(n, dim=19, n_snakes=NS, seed=0)
env.reset_device()
tape = torch.from_numpy(np.random.default_rng(1234).integers(0, 5, (T, n, NS)).astype(np.int32)).cuda()
L, h = env._L, env._h
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(m):
    k = 0
    while k < m:
        c = min(T, m - k)
        msnake._capi.check(L.msnake_step_tape(h, tape.data_ptr(), NS, c, env._obs.data_ptr(), 0, env._rew.data_ptr(),
                                              env._done.data_ptr(), env._info.data_ptr(), 0, st), "step_tape")
        k += c


run(16)
torch.cuda.synchronize()
out = []
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(96); e1.record()
    torch.cuda.synchronize()
    out.append(round(e0.elapsed_time(e1) * 1e3 / 96, 1))
print(mode, os.path.basename(msnake._capi.LIB_PATH), out, hex(env._obs.data_ptr()))
