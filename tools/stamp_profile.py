#!/usr/bin/env python3
"""Diagnostic (MSNAKE_DBG_STAGES build only): per-wave wall-clock stamps of one steady-state step.
Prints, per stage, when the fastest / median / slowest wave reached it (us since the first wave
started) split by which slow path the wave took."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
buf = torch.zeros((2 * n, 8), dtype=torch.int64, device="cuda")
os.environ["MSNAKE_DBG_BUF"] = hex(buf.data_ptr())
import msnake

env = msnake.MultiSnakeVecEnv(n, dim=19, n_snakes=3, seed=0)
env.reset_device()
tape = torch.randint(0, 5, (300, n, 3), dtype=torch.int32, device="cuda")
for t in range(200):
    env.step_device(tape[t])
names = ["entry", "loads", "move", "alive", "reset+stats", "painted", "stores issued"]
acc = {}
for rep in range(20):
    torch.cuda.synchronize()
    env.step_device(tape[200 + rep])
    torch.cuda.synchronize()
    ball = buf.cpu().numpy().astype(np.int64)
    b, b2 = ball[:n], ball[n:]
    eat = (b[:, 7] & 1) == 1
    if eat.any():  # inside the respawn path (second stamp row): since the move section began
        d2 = np.stack([(b2[eat, 0] - b[eat, 1]), (b2[eat, 1] - b2[eat, 0]), (b2[eat, 2] - b2[eat, 1]), (b2[eat, 3] - b2[eat, 2]), (b[eat, 2] - b2[eat, 3])], 1) / 100.0
        acc.setdefault("eat_split", []).append(np.median(d2, 0))
    t0 = b[:, 0].min()
    us = (b[:, :7] - t0) / 100.0
    flag = b[:, 7]
    for cls, sel in (("all", np.ones(n, bool)), ("fast", flag == 0), ("eat", (flag & 1) == 1), ("done", (flag & 2) == 2)):
        if sel.sum() == 0:
            continue
        acc.setdefault(cls, []).append(np.stack([us[sel].min(0), np.median(us[sel], 0), us[sel].max(0)]))
        acc.setdefault(cls + "_n", []).append(sel.sum())
dur = np.diff(us, axis=1)  # of the last repeat: per-wave time spent between consecutive stamps
print("--- per-wave stage durations (last repeat): median / p90 / max us, and whole wave life")
for i in range(6):
    print(f"  {names[i]:>14s} -> {names[i + 1]:14s} {np.median(dur[:, i]):6.2f} {np.percentile(dur[:, i], 90):6.2f} {dur[:, i].max():6.2f}")
life = us[:, 6] - us[:, 0]
print(f"  wave life      median {np.median(life):.2f}  p90 {np.percentile(life, 90):.2f}  max {life.max():.2f};  last wave start {us[:, 0].max():.2f}, last stores issued {us[:, 6].max():.2f}")
order = np.argsort(us[:, 0])
late = order[-256:]
print(f"  the 256 waves that start last: life median {np.median(life[late]):.2f}; stage medians " + " ".join(f"{np.median(dur[late, i]):.2f}" for i in range(6)))
if "eat_split" in acc:
    m = np.mean(acc["eat_split"], 0)
    print("--- respawn path of the waves that eat, median us: vector prologue %.2f, draws %.2f, free-cell map %.2f, pick + record %.2f, rest of the move section %.2f" % tuple(m))
for cls in ("all", "fast", "eat", "done"):
    if cls not in acc:
        continue
    m = np.mean(acc[cls], 0)
    print(f"--- {cls} waves (avg {np.mean(acc[cls + '_n']):.0f} of {n}): min / median / max us since first wave start")
    for i, nm in enumerate(names):
        print(f"  {nm:14s} {m[0, i]:6.2f} {m[1, i]:6.2f} {m[2, i]:6.2f}")
