#!/usr/bin/env python
"""Checker for the hand-waited LDS reads of scripts/micro/wino_b3v2.hip: in the ISA listing (hipcc -S --cuda-device-only), between an asm
ds_read2st64_b64 into a register tuple and the s_waitcnt that completes it (in-order LDS queue: lgkmcnt(n) leaves the newest n reads
outstanding), NO instruction may touch those registers -- the compiler does not know they are pending and is free to put a copy there.

    python scripts/micro/check_pending_reads.py listing.s [kernel-name-prefix]"""
import re
import sys

txt = open(sys.argv[1]).read()
prefix = sys.argv[2] if len(sys.argv) > 2 else "_Z16conv_wino_res_b3"
bad = 0
HAND = ("ds_read2st64_b64", "ds_read_b64")     # the kernel's asm reads (the compiler's own are ds_read_b128 and waited for by it)
for k in re.split(r"\n(?=_Z\w+:)", txt):
    if not k.startswith(prefix):
        continue
    name = k.split(":")[0]
    queue = []          # outstanding LDS reads in issue order: sets of destination registers (None: a read whose registers the compiler tracks)
    nreads = 0
    for ln, line in enumerate(k.split("\n")):
        line = line.split(";")[0].strip()
        if not line or line.endswith(":") or line.startswith("."):
            continue
        op = line.split()[0]
        regs = set()
        for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", line):
            regs.update(range(int(m.group(1)), int(m.group(2)) + 1) if m.group(1) else [int(m.group(3))])
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", line)
            if m:
                n = int(m.group(1))
                queue = queue[len(queue) - n:] if n else []
            continue
        if op.startswith("s_cbranch") or op in ("s_branch", "s_barrier", "s_endpgm"):
            continue
        pending = set().union(*[q for q in queue if q]) if queue else set()
        hit = regs & pending
        if op.startswith("ds_read"):
            m = re.match(r"\S+\s+v\[(\d+):(\d+)\]", line)
            dst = set(range(int(m.group(1)), int(m.group(2)) + 1)) if m else set()
            src_hit = (regs - dst) & pending
            if src_hit or (dst & pending):
                print("%s line %d: %s touches pending %s" % (name, ln, line, sorted(src_hit | (dst & pending))))
                bad += 1
            queue.append(dst if op in HAND else None)
            nreads += op in HAND
            continue
        if op.startswith("ds_") or op.startswith("s_load") or op.startswith("s_buffer_load"):
            queue.append(None)
        if hit:
            print("%s line %d: %s touches pending %s" % (name, ln, line, sorted(hit)))
            bad += 1
    print("%s: %d hand-waited reads checked" % (name, nreads))
print("violations: %d" % bad)
sys.exit(1 if bad else 0)
