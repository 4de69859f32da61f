#!/usr/bin/env python3
"""Instruction histogram per kernel from a hipcc -S dump (tools/isa_stats.py file.s [filter])."""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for f in re.split(r'\n(?=_ZN6msnake18msnake_step_kernel\w+:)', s)[1:]:
    name = f.split(':')[0]
    if flt not in name:
        continue
    body = f.split('s_endpgm')[0]
    lines = [l.strip() for l in body.split('\n') if l.strip() and not l.strip().startswith(('.', ';', '/'))]
    c = Counter(l.split()[0] for l in lines)
    keys = ['v_readlane_b32', 'v_writelane_b32', 'ds_write_b8', 'ds_write_b128', 'ds_read_b128', 'global_load_dwordx4',
            'global_store_dwordx4', 'global_load_ushort', 's_waitcnt', 'scratch_load_dword', 'scratch_store_dword']
    print(name, 'instrs', len(lines), {k: c[k] for k in keys if c.get(k)})
