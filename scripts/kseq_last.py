#!/usr/bin/env python
"""Kernel sequence of the LAST step of a rocprofv3 --kernel-trace .db, delimited by a marker kernel (default: adam_step_kernel, the
last kernel of a training step) -- for traces whose kernel count per step is not constant (eager warm-up steps, data parallel).
    kseq_last.py results.db [marker]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else "adam_step_kernel"
rows = list(db.execute("select name, start, end from kernels order by start"))
ends = [i for i, r in enumerate(rows) if r[0].startswith(marker)]
last = rows[ends[-2] + 1: ends[-1] + 1]
t0, prev = last[0][1], last[0][1]
for name, s, e in last:
    print("%9.1f %8.1f %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, name[:100]))
    prev = max(prev, e)
print("step span %.1f us, %d kernels, kernel time %.1f us" % ((last[-1][2] - t0) / 1e3, len(last), sum(e - s for _, s, e in last) / 1e3))
