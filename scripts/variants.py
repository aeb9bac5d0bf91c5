#!/usr/bin/env python
"""Build VARIANT libraries for A/B experiments without touching the shipped one (round-4 advice): each variant = the shipped objects
with ONE source recompiled under extra -D flags, linked to csrc/build/variants/libaesr_<name>.so; a process loads it with
AESR_LIB=<path> (superresolution_aniso_mri_amd/_hip.py says so on stderr).  Run `make -C superresolution_aniso_mri_amd/csrc` first.

    python scripts/variants.py name=file.hip:-DA=1,-DB=2 [name2=...]
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "superresolution_aniso_mri_amd", "csrc")
VAR = os.path.join(CS, "build", "variants")
FLAGS = ["-O3", "-fPIC", "-std=c++17", "--offload-arch=gfx950", "-Wall", "-Wno-unused-result", "-ffp-contract=off"]
NOSLP = ("conv_wgrad_wino.hip", "conv_wino.hip", "conv_wino_res.hip", "conv_wino_ring.hip")


def build(name, src, defs):
    os.makedirs(VAR, exist_ok=True)
    obj = os.path.join(VAR, "%s_%s.o" % (name, src.replace(".hip", "")))
    flags = FLAGS + (["-fno-slp-vectorize"] if src in NOSLP else []) + defs
    subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-c", os.path.join(CS, src), "-o", obj], check=True)
    objs = [obj if o == src.replace(".hip", ".o") else os.path.join(CS, "build", o)
            for o in sorted(f for f in os.listdir(os.path.join(CS, "build")) if f.endswith(".o"))]
    out = os.path.join(VAR, "libaesr_%s.so" % name)
    subprocess.run(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", out] + objs + ["-ldl"], check=True)
    print(out)


if __name__ == "__main__":
    from concurrent.futures import ThreadPoolExecutor
    jobs = []
    for spec in sys.argv[1:]:
        name, rest = spec.split("=", 1)
        src, _, defs = rest.partition(":")
        jobs.append((name, src, [d for d in defs.split(",") if d]))
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(lambda j: build(*j), jobs))
