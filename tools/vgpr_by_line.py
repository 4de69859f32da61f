#!/usr/bin/env python3
"""Diagnostic: highest VGPR index touched per source line of one step-kernel instantiation.
usage: tools/vgpr_by_line.py [mangled-name-substring] [min-index]   (compiles csrc/msnake_kernels.hip -S)"""
import os
import re
import subprocess
import sys
from collections import defaultdict

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "self-play-on-multi-snakes-environment_amd", "csrc", "msnake_kernels.hip")
key = sys.argv[1] if len(sys.argv) > 1 else "ILi0ELi3ELi0ELi1E"
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 28
out = "/tmp/msnake_vgpr.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-Os", "-std=c++17", "--offload-arch=gfx950", "-mllvm",
                       "-amdgpu-kernarg-preload-count=16", "-gline-tables-only", "-S", "--cuda-device-only",
                       "-o", out, src], stderr=subprocess.DEVNULL)
s = open(out).read()
i = s.index("\n_ZN6msnake18msnake_step_kernel" + key)
body = s[i:s.index("s_endpgm", i)]
cur, hi, cnt = None, defaultdict(int), defaultdict(int)
for l in body.split("\n"):
    m = re.search(r"\.loc\s+\d+\s+(\d+)", l)
    if m:
        cur = int(m.group(1))
        continue
    l = l.split(";")[0]
    regs = [int(x) for x in re.findall(r"\bv(\d+)\b", l)] + [int(b) for _, b in re.findall(r"v\[(\d+):(\d+)\]", l)]
    if regs:
        hi[cur] = max(hi[cur], max(regs))
        cnt[cur] += 1
for ln, v in sorted(hi.items()):
    if v >= lo:
        print(f"line {ln}: v{v} ({cnt[ln]} instrs)")
