#!/usr/bin/env python
"""Per-kernel means of the PMC counters over the LAST 30 dispatches of each kernel (the N = 36 timing loop of wino_b3v2.py)."""
import csv
import glob
import sys
from collections import defaultdict

out = sys.argv[1]
for p in ("p1", "p2"):
    files = glob.glob("%s/%s/**/*counter_collection.csv" % (out, p), recursive=True)
    if not files:
        print(p, "no counter file")
        continue
    rows = defaultdict(lambda: defaultdict(list))       # kernel -> counter -> values in dispatch order
    for r in csv.DictReader(open(files[0])):
        k = r["Kernel_Name"].split("(")[0][:40]
        if "wino_res" not in k:
            continue
        rows[k][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for k, cs in rows.items():
        print("%s  %s" % (p, k))
        for c, vals in sorted(cs.items()):
            vals = [v for _, v in sorted(vals)][-30:]
            print("    %-28s %16.0f" % (c, sum(vals) / len(vals)))
