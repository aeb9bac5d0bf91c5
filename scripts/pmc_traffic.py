#!/usr/bin/env python
"""HBM-side traffic per kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same command).
pmc_traffic.py fetch_counter_collection.csv write_counter_collection.csv steps out.json
Corrections (MI355X_MICROARCH.md, HBM section): the counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of
wide (16 B/lane) coalesced reads -> doubled; WRITE_SIZE is exact for 16 B/lane streaming stores."""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (csrc_sha: the kernel sources these passes were measured on; bench.py quotes the file only while they match)


def load(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0]
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    return agg


f = load(sys.argv[1], "FETCH_SIZE")
w = load(sys.argv[2], "WRITE_SIZE")
steps = int(sys.argv[3])
out = {}
for k in sorted(set(f) | set(w), key=lambda k: -(f.get(k, [0, 0])[1] * 2 + w.get(k, [0, 0])[1])):
    nf, fb = f.get(k, [0, 0.0])
    nw, wb = w.get(k, [0, 0.0])
    n = max(nf, nw)
    out[k] = {"launches_per_step": n / steps, "read_MB_per_launch": round(2 * fb * 1024 / max(n, 1) / 1e6, 3),
              "write_MB_per_launch": round(wb * 1024 / max(n, 1) / 1e6, 3),
              "MB_per_step": round((2 * fb + wb) * 1024 / steps / 1e6, 1)}
tot = sum(v["MB_per_step"] for v in out.values())
res = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KiB units, FETCH_SIZE doubled per the gfx950 correction; "
               "memory-side requests of the L2 (Infinity-Cache hits are included)", "steps": steps, "csrc_sha": bench.csrc_sha(),
       "total_MB_per_step": round(tot, 1), "kernels": out}
json.dump(res, open(sys.argv[4], "w"), indent=1)
print("total %.1f MB/step" % tot)
for k, v in list(out.items())[:16]:
    print("%-60s %5.1f/step  rd %8.2f MB  wr %8.2f MB  per launch;  %8.1f MB/step" % (k[:60], v["launches_per_step"], v["read_MB_per_launch"], v["write_MB_per_launch"], v["MB_per_step"]))
