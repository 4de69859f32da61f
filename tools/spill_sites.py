#!/usr/bin/env python3
"""Where does the compiler spill SGPRs (v_writelane_b32 into a spill VGPR) in one kernel instantiation?
usage: spill_sites.py file.s(-gline-tables-only) source.hip mangled-prefix"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
src = open(sys.argv[2]).read().split("\n")
f = [x for x in re.split(r"\n(?=_ZN6msnake18msnake_step_kernel\w+:)", s)[1:] if x.startswith(sys.argv[3])][0]
cur, wl, in_asm = 0, collections.Counter(), False
for l in f.split("s_endpgm")[0].split("\n"):
    t = l.strip()
    m = re.match(r"\.loc\s+\d+\s+(\d+)\s+(\d+)", t)
    if m:
        cur = int(m.group(1))
    elif "#ASMSTART" in t:
        in_asm = True
    elif "#ASMEND" in t:
        in_asm = False
    elif "v_writelane_b32" in t and not in_asm:
        wl[cur] += 1
for line, c in sorted(wl.items()):
    print(line, c, src[line - 1].strip()[:110] if line else "")
