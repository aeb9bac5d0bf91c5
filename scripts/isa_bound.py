#!/usr/bin/env python
"""Instruction-count bound of a Winograd MFMA kernel's item loop from its ISA (hipcc -S): what the matrix pipe can reach at most when every
non-MFMA instruction costs what scripts/micro/mfma_gap.hip measured on gfx950 (profiles/r02_micro_mfma_gap.txt): an f32 MFMA holds its SIMD
32 cycles and NOTHING a wave (or its SIMD partner) issues beside it is free -- a vector ALU instruction 4-5 cycles (4.0-4.3 with one wave per
SIMD, 4.7-6.5 with two), an LDS-DMA ~63, a 16-byte store ~63 (assumed = DMA), an LDS read ~1, s_waitcnt / s_nop ~2.

    python scripts/isa_bound.py file.s mangled_kernel_name LOOPHEADER      (blocks "in Loop: Header=<LOOPHEADER>" + the header itself)
"""
import re
import sys
from collections import Counter

COST = {"mfma": 32.0, "valu": 4.5, "vpk": 4.5, "dma": 63.0, "store": 63.0, "ds": 1.0, "wait": 2.0, "nop": 2.0, "salu": 0.0, "load": 20.0, "other": 0.0}


def cls(op, line):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("ds_"): return "ds"
    if op.startswith("buffer_load") and " lds" in line: return "dma"
    if op.startswith(("buffer_load", "global_load")): return "load"
    if op.startswith(("buffer_store", "global_store")): return "store"
    if op.startswith("v_pk"): return "vpk"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_"): return "salu"
    return "other"


s = open(sys.argv[1]).read()
name, header = sys.argv[2], sys.argv[3]
body = re.search(re.escape(name) + r":(.*?)\.Lfunc_end", s, re.S).group(1)
blocks, cur, nm = [], [], "entry"
for l in body.split("\n"):
    t = l.strip()
    if not t or t.startswith(";"):
        continue
    if re.match(r"^\.LBB\d+_\d+:", t):
        blocks.append((nm, cur)); nm, cur = t, []
        continue
    if t.startswith("."):
        continue
    cur.append(t)
blocks.append((nm, cur))
tot = Counter()
print("%-12s %5s  %s" % ("block", "instr", "mix"))
for nm, b in blocks:
    if ("Header=" + header) not in nm and not nm.startswith("." + header.replace("BB", "LBB") + ":"):
        continue
    c = Counter(cls(l.split()[0], l) for l in b)
    # the sigmoid blocks (v_exp / v_rcp / v_div_*) are skipped by a branch in the layers of the auto-encoder's 3x3 stack: not on the path
    if any(l.split()[0].startswith(("v_exp", "v_div_", "v_rcp")) for l in b):
        keep = [l for l in b if not re.match(r"v_(exp|div_|rcp|rndne|ldexp|cvt_i32|cmp_n[lg]t)", l.split()[0])]
        print("%-12s %5d  (sigmoid arm, branched over: %d of its instructions not counted)" % (nm.split(":")[0], len(b), len(b) - 20))
        c = Counter({"valu": 20, "store": c["store"]})
    else:
        print("%-12s %5d  %s" % (nm.split(":")[0], len(b), dict(c)))
    tot.update(c)
cyc = {k: v * COST[k] for k, v in tot.items()}
total = sum(cyc.values())
print("\nper pass over the loop (one item = all its chunks + epilogue; branches not taken are counted once):")
for k, v in sorted(cyc.items(), key=lambda kv: -kv[1]):
    if v:
        print("   %-6s %5d x %4.1f = %7.0f cycles (%4.1f %%)" % (k, tot[k], COST[k], v, 100 * v / total))
print("   bound on the matrix pipe: %.0f / %.0f = %.1f %%" % (cyc.get("mfma", 0), total, 100 * cyc.get("mfma", 0) / total))
