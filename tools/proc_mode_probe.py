#!/usr/bin/env python3
"""What distinguishes a 'slow' process (6.0-6.15 us per 4 096-env step) from a 'fast' one (5.7)?  One process =
one sample: the step time (HIP events, 1 024 launches, median of 5), the latency of the process's FIRST launch, the
shader / memory clocks rocm-smi reports before and after the timed region, and which HW queue the stream got (the
doorbell / queue id is not exposed through HIP: the stream handle and the order of creation stand in).  Run it N
times in a row (tools/proc_mode.sh) and compare the columns.  Prints one JSON line."""
import ctypes
import json
import os
import re
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def smi():
    """current sclk / mclk / power of GPU 0 as rocm-smi prints them (a child process: it does not touch our queue)"""
    try:
        txt = subprocess.run(["rocm-smi", "-d", "0", "-c", "-P", "--showperflevel"], capture_output=True, text=True, timeout=20).stdout
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)[:80]}
    out = {}
    for key, pat in (("sclk_mhz", r"sclk clock level: \d+: \((\d+)Mhz\)"), ("mclk_mhz", r"mclk clock level: \d+: \((\d+)Mhz\)"),
                     ("fclk_mhz", r"fclk clock level: \d+: \((\d+)Mhz\)"), ("socclk_mhz", r"socclk clock level: \d+: \((\d+)Mhz\)"),
                     ("power_w", r"Power \(W\): ([\d.]+)"), ("perf_level", r"Performance Level: (\w+)")):
        m = re.search(pat, txt)
        if m:
            out[key] = m.group(1)
    return out


before = smi()
import numpy as np
import torch
import msnake

n, NS, K = 4096, 3, 1024
t0 = time.perf_counter()
env = msnake.MultiSnakeVecEnv(n, dim=19, n_snakes=NS, seed=0, device="cuda:0")
t_create = time.perf_counter() - t0
tape = torch.from_numpy(np.random.default_rng(1234).integers(0, 5, (256, n, NS)).astype(np.int32)).cuda()
torch.cuda.synchronize()
t0 = time.perf_counter()
env.reset_device()
torch.cuda.synchronize()
first_launch_us = (time.perf_counter() - t0) * 1e6
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
reps = []
for _ in range(5):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(K); e1.record()
    torch.cuda.synchronize()
    reps.append(e0.elapsed_time(e1) * 1e3 / K)
during = smi()
# an empty-ish kernel's cadence on the same stream: is the whole queue slow, or this kernel?
x = torch.zeros(64, device="cuda")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(2000):
    x.add_(1.0)
e1.record()
torch.cuda.synchronize()
tiny_us = e0.elapsed_time(e1) * 1e3 / 2000
after = smi()
print(json.dumps({"us_per_step": round(sorted(reps)[2], 3), "repeats": [round(r, 3) for r in reps], "first_launch_us": round(first_launch_us, 1),
                  "create_ms": round(t_create * 1e3, 1), "tiny_kernel_us": round(tiny_us, 3), "stream_handle": hex(st.value or 0),
                  "state_ptr_low": hex(env._obs.data_ptr() & 0xFFFFFF), "obs_ptr": hex(env._obs.data_ptr()), "pid": os.getpid(),
                  "smi_before": before, "smi_during": during, "smi_after": after}))
