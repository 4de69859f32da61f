#!/usr/bin/env python3
"""Static instruction count per source line of one kernel (hipcc -S -gline-tables-only dump).
usage: isa_by_line.py file.s source.hip mangled-prefix [min_count]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
src = open(sys.argv[2]).read().split('\n')
prefix = sys.argv[3]
minc = int(sys.argv[4]) if len(sys.argv) > 4 else 6
f = [x for x in re.split(r'\n(?=_ZN6msnake18msnake_step_kernel\w+:)', s)[1:] if x.startswith(prefix)][0]
body = f.split('s_endpgm')[0]
cur = 0
hist = collections.Counter()
kinds = collections.defaultdict(collections.Counter)
for l in body.split('\n'):
    t = l.strip()
    m = re.match(r'\.loc\s+\d+\s+(\d+)\s+(\d+)', t)
    if m:
        cur = int(m.group(1))
        continue
    if not t or t.startswith(('.', ';', '/')) or t.endswith(':'):
        continue
    op = t.split()[0]
    hist[cur] += 1
    kinds[cur]['S' if op.startswith('s_') else 'V' if op.startswith('v_') else 'M'] += 1
print('total static', sum(hist.values()))
for line, c in sorted(hist.items()):
    if c >= minc:
        print(f"{line:4d} {c:4d} S{kinds[line]['S']:3d} V{kinds[line]['V']:3d} M{kinds[line]['M']:3d} | {src[line - 1].strip()[:110] if line else ''}")
