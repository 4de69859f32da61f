#!/usr/bin/env python3
"""Is the slow process mode a host-placement effect?  The GPU reads every launch's AQL packet and kernel arguments
from memory the HIP runtime allocated where the process was running; on a two-socket host that memory is one or two
fabric hops from the GPU.  This probe pins the process (and therefore its first-touch allocations) to one NUMA node
BEFORE anything touches the GPU, then times 1 024 back-to-back launches like tools/proc_mode_probe.py.
usage: numa_mode_probe.py <node | none>     prints one JSON line"""
import glob
import json
import os
import sys


def cpulist(path):
    out = []
    for part in open(path).read().strip().split(","):
        if "-" in part:
            a, b = part.split("-"); out += list(range(int(a), int(b) + 1))
        elif part:
            out.append(int(part))
    return out


nodes = {int(os.path.basename(p)[4:]): cpulist(p + "/cpulist") for p in glob.glob("/sys/devices/system/node/node[0-9]*")}
gpu_nodes = {}
for p in glob.glob("/sys/class/drm/card*/device/numa_node"):
    try:
        gpu_nodes[p.split("/")[4]] = int(open(p).read())
    except Exception:  # noqa: BLE001
        pass
want = sys.argv[1] if len(sys.argv) > 1 else "none"
allowed = sorted(os.sched_getaffinity(0))
if want != "none":
    cpus = [c for c in nodes[int(want)] if c in allowed]
    os.sched_setaffinity(0, cpus)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import numpy as np
import torch
import msnake

n, NS, K = 4096, 3, 1024
env = msnake.MultiSnakeVecEnv(n, dim=19, n_snakes=NS, seed=0, device="cuda:0")
tape = torch.from_numpy(np.random.default_rng(1234).integers(0, 5, (256, n, NS)).astype(np.int32)).cuda()
env.reset_device()
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
# where do this process's kernel arguments live?  (tools/probes/kernarg_probe.hip)
ka = {}
try:
    KP = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes", "libkernarg_probe.so"))
    KP.kernarg_probe.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    out = torch.zeros((32, 4), dtype=torch.int64, device="cuda")
    KP.kernarg_probe(out.data_ptr(), 32, st)
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    assert (o[:, 3] == 7000 + np.arange(32)).all()
    ka = {"kernarg_load_ns": int(np.median(o[8:, 0]) * 10),  # (the compiler issues the argument load at kernel entry: the first wait covers it)
          "kernarg_ptr": hex(int(o[8, 2])), "obs_ptr": hex(env._obs.data_ptr())}
    KP.xcc_probe.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    xs = []
    for _ in range(3):
        xo = torch.full((16,), -1, dtype=torch.int64, device="cuda")
        KP.xcc_probe(xo.data_ptr(), 16, st)
        torch.cuda.synchronize()
        xs.append("".join(str(int(v)) for v in xo.cpu().tolist()))
    ka["xcc_of_blocks_0_15"] = xs
except Exception as e:  # noqa: BLE001
    ka = {"error": str(e)[:100]}
cpu_now = None
try:
    cpu_now = int(open("/proc/self/stat").read().split()[38])
except Exception:  # noqa: BLE001
    pass
node_now = [k for k, v in nodes.items() if cpu_now in v]
print(json.dumps({"pinned_node": want, "us_per_step": round(sorted(reps)[2], 3), "running_on_cpu": cpu_now, "running_on_node": node_now,
                  "gpu_numa_nodes": gpu_nodes, "n_nodes": len(nodes), "allowed_cpus": len(allowed),
                  "HIP_FORCE_DEV_KERNARG": os.environ.get("HIP_FORCE_DEV_KERNARG"), "kernarg": ka}))
