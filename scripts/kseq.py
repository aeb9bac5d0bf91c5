#!/usr/bin/env python
"""Kernel SEQUENCE of the last step in a rocprofv3 --kernel-trace results .db: kseq.py results.db steps
(start offset, duration and the gap to the previous kernel, us) -- shows what sits between the big kernels"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2])
rows = list(db.execute("select name, start, end from kernels order by start"))
per = len(rows) // steps
last = rows[-per:]
t0 = last[0][1]
prev_end = t0
gap_total = 0.0
for name, s, e in last:
    gap = (s - prev_end) / 1e3
    gap_total += max(gap, 0.0)
    print("%9.1f %8.1f %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, name[:100]))
    prev_end = max(prev_end, e)
print("step span %.1f us, %d kernels, idle gaps %.1f us" % ((last[-1][2] - t0) / 1e3, len(last), gap_total))
