#!/usr/bin/env python3
"""Probe: CnnPolicy forward / forward+backward time at B=1024 chunks, fp32 vs bf16 autocast (MIOpen)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
import torch
import msnake
from msnake import selfplay

dev = torch.device("cuda", 0)
for shape, B in (((21, 21, 3), 8192), ((84, 84, 3), 2048)):
    model = selfplay.CnnPolicy(shape).to(dev)
    x = torch.randint(0, 256, (B,) + shape, dtype=torch.uint8, device=dev)
    for name, ctx in (("fp32", torch.autocast("cuda", enabled=False)), ("bf16 autocast", torch.autocast("cuda", dtype=torch.bfloat16)),
                      ("fp16 autocast", torch.autocast("cuda", dtype=torch.float16))):
        for mode in ("fwd", "fwd+bwd"):
            ts = []
            for rep in range(4):
                torch.cuda.synchronize(); t0 = time.time()
                with ctx:
                    if mode == "fwd":
                        with torch.no_grad():
                            lg, v = model(x)
                    else:
                        lg, v = model(x)
                        loss = lg.float().square().mean() + v.float().square().mean()
                if mode != "fwd":
                    model.zero_grad(); loss.backward()
                torch.cuda.synchronize(); ts.append(time.time() - t0)
            print(f"{shape} B={B} {name:14s} {mode:8s}: {min(ts)*1e3:8.2f} ms  ({B/min(ts)/1e3:.0f} K frames/s)", flush=True)
