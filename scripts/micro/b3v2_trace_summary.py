#!/usr/bin/env python
"""Kernel durations of wino_b3v2.py's timing loops from a rocprofv3 --kernel-trace CSV: per batch size (4, 12, 36 images) the mean of the last 30
launches of the block of launches the driver makes for it (41 of the b3 kernel: 1 + 5 equality runs + 5 warm-up + 30 timed; 36 of the library's)."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
b3 = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "conv_wino_res_b3" in r["Kernel_Name"]]
lib = [((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"].split("(")[0]) for r in rows if "conv_wino_res_f32" in r["Kernel_Name"]]
b3 = b3[len(b3) - 3 * 41:]
assert len(lib) == 3 * 36, len(lib)
print("kernel durations (rocprofv3 --kernel-trace), 32 -> 32 channels @ 160 x 160, LeakyReLU forward, mean of 30 launches")
for i, n in enumerate((4, 12, 36)):
    tb = b3[41 * i:41 * (i + 1)][-30:]
    tl = lib[36 * i:36 * (i + 1)][-30:]
    mb, ml = sum(tb) / 30, sum(t for t, _ in tl) / 30
    print("  %2d images: conv_wino_res_b3 %.1f us | %s %.1f us | ratio %.2f" % (n, mb, tl[0][1].replace("void ", ""), ml, ml / mb))
