#!/usr/bin/env python
"""Per-kernel summary of a rocprofv3 --kernel-trace results .db: kstats.py results.db [steps]  (us per step, avg/min/max)"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = list(db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
                       "group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print("total kernel time %.3f ms/step over %d steps" % (tot / steps / 1e6, steps))
print("%-96s %6s %10s %9s %9s %9s" % ("kernel", "calls", "us/step", "avg us", "min us", "max us"))
for r in rows:
    print("%-96s %6d %10.1f %9.1f %9.1f %9.1f" % (r[0][:96], r[1], r[2] / steps / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3))
