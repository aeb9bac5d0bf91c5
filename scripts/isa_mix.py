#!/usr/bin/env python
"""Instruction mix of one kernel in a hipcc -S listing:  python scripts/isa_mix.py file.s mangled_kernel_name [--loop]"""
import re
import sys
from collections import Counter


def mix(lines):
    c = Counter()
    for l in lines:
        op = l.split()[0]
        if op.startswith('v_mfma'): c['mfma'] += 1
        elif op.startswith(('ds_read', 'ds_load')): c['ds_read'] += 1
        elif op.startswith(('ds_write', 'ds_store')): c['ds_write'] += 1
        elif op.startswith('buffer_load'): c['buffer_load'] += 1
        elif op.startswith('buffer_store'): c['buffer_store'] += 1
        elif op.startswith('v_'): c['valu'] += 1
        elif op.startswith('s_waitcnt'): c['waitcnt'] += 1
        elif op.startswith('s_barrier'): c['barrier'] += 1
        elif op.startswith('s_nop'): c['s_nop'] += 1
        elif op.startswith('s_'): c['salu'] += 1
        else: c[op] += 1
    return c


s = open(sys.argv[1]).read()
m = re.search(re.escape(sys.argv[2]) + r':(.*?)\.Lfunc_end', s, re.S)
body = m.group(1)
blocks, cur, name = [], [], "entry"
for l in body.split('\n'):
    t = l.strip()
    if not t or t.startswith(';'):
        continue
    if re.match(r'^\.LBB\d+_\d+:', t):
        blocks.append((name, cur)); name, cur = t, []
        continue
    if t.startswith('.'):
        continue
    cur.append(t)
blocks.append((name, cur))
tot = [l for _, b in blocks for l in b]
print(len(tot), "instructions total", dict(mix(tot)))
for name, b in blocks:
    c = mix(b)
    if c['mfma'] >= 16:
        print(name, len(b), dict(c))
