#!/usr/bin/env python
"""Per-kernel statistics of the LAST K steps of a rocprofv3 --kernel-trace .db, a step ending with the marker kernel (default
adam_step_kernel): for traces of REPLAYED graphs, whose first steps (eager warm-up, capture) have other kernel counts.
    kstats_last.py results.db [K] [marker] [out.json]  -> us per step per kernel, launches per step, and the span of the steps; with out.json the
    same numbers machine-readable, stamped with the hash of the kernel sources (bench.csrc_sha): bench.py quotes ``roofline.rocprof`` and the
    tail's times from the newest such file whose hash equals the sources it runs on"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
marker = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] else "adam_step_kernel"
rows = list(db.execute("select name, start, end from kernels order by start"))
ends = [i for i, r in enumerate(rows) if r[0].startswith(marker)]
K = min(K, len(ends) - 1)
sel = rows[ends[-K - 1] + 1: ends[-1] + 1]
agg = collections.OrderedDict()
for name, s, e in sel:
    d = agg.setdefault(name, [0, 0.0, 1e30, 0.0])
    d[0] += 1
    d[1] += (e - s) / 1e3
    d[2] = min(d[2], (e - s) / 1e3)
    d[3] = max(d[3], (e - s) / 1e3)
span = (sel[-1][2] - sel[0][1]) / 1e3 / K
tot = sum(v[1] for v in agg.values()) / K
print("last %d steps (delimited by %s): %.1f us of kernel time per step, %.1f us from the first kernel's start to the last kernel's end per step, %d kernels per step"
      % (K, marker, tot, span, len(sel) // K))
print("%-100s %7s %10s %9s %9s %9s" % ("kernel", "n/step", "us/step", "avg us", "min us", "max us"))
for name, (n, t, lo, hi) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-100s %7.1f %10.1f %9.1f %9.1f %9.1f" % (name[:100], n / K, t / K, t / n, lo, hi))
if len(sys.argv) > 4:
    import json
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench  # noqa: E402
    json.dump({"note": "rocprofv3 --kernel-trace of the REPLAYED step, last %d steps (scripts/kstats_last.py)" % K, "csrc_sha": bench.csrc_sha(), "steps": K,
               "kernel_us_per_step": round(tot, 1), "span_us_per_step": round(span, 1), "kernels_per_step": len(sel) // K,
               "kernels": {name: {"n_per_step": round(n / K, 2), "us_per_step": round(t / K, 2), "avg_us": round(t / n, 2)}
                           for name, (n, t, lo, hi) in sorted(agg.items(), key=lambda kv: -kv[1][1])}}, open(sys.argv[4], "w"), indent=1)
