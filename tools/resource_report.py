#!/usr/bin/env python3
"""Per-kernel register / spill table from `make -C <pkg>/csrc report` (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/resource_report.py [filter]   -- prints name, SGPRs, VGPRs, scratch, SGPR spills, VGPR spills"""
import os
import re
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(root, "self-play-on-multi-snakes-environment_amd", "csrc")
out = subprocess.run(["make", "-C", csrc, "report"], capture_output=True, text=True).stderr
flt = sys.argv[1] if len(sys.argv) > 1 else ""
rows, cur = [], None
for line in out.split("\n"):
    m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|SGPRs Spill|VGPRs Spill): (\S+)", line)
    if not m:
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    else:
        cur[k] = v
demangle = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.split("\n")
print(f"{'kernel':44s} {'SGPR':>5s} {'VGPR':>5s} {'scratch':>8s} {'sSpill':>7s} {'vSpill':>7s}")
worst = 0
for r, d in zip(rows, demangle):
    short = re.sub(r"^void msnake::|\(.*$", "", d)
    if flt and flt not in short:
        continue
    print(f"{short:44s} {r['TotalSGPRs']:>5s} {r['VGPRs']:>5s} {r['ScratchSize [bytes/lane]']:>8s} {r['SGPRs Spill']:>7s} {r['VGPRs Spill']:>7s}")
    worst = max(worst, int(r["SGPRs Spill"]))
print("max SGPR spill:", worst)
