#!/usr/bin/env python
"""Build the timing-only ablation variants of conv_wino_res_f32 that scripts/r04_res_ablation.sh times (results wrong by construction):
a scratch copy of csrc/conv_wino_res.hip with #if WR_ABL guards, compiled with -DWR_ABL=0..5 into csrc/build/variants/libaesr_resabl<n>.so
(the other objects are the shipped ones: run `make -C superresolution_aniso_mri_amd/csrc` first).
   0 as shipped | 1 no patch DMAs | 2 no input transform | 3 no LDS reads of patch / filter inside the chunk loop | 4 MFMAs only (1 + 2 + 3 + 5) | 5 no epilogue"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "superresolution_aniso_mri_amd", "csrc")
VAR = os.path.join(CS, "build", "variants")
os.makedirs(VAR, exist_ok=True)
s = open(os.path.join(CS, "conv_wino_res.hip")).read()


def sub(old, new):
    global s
    if old not in s:
        sys.exit("ablation patch no longer matches conv_wino_res.hip: %r" % old[:70])
    s = s.replace(old, new, 1)


sub('#include "aesr_kernels.h"', '#ifndef WR_ABL\n#define WR_ABL 0\n#endif\n#include "aesr_kernels.h"')
sub("            wr_dma(rs_in, ldsP + r * WR_RP, off + urow);",
    "#if WR_ABL != 1 && WR_ABL != 4\n            wr_dma(rs_in, ldsP + r * WR_RP, off + urow);\n#else\n"
    "            if (cc == 99) wr_dma(rs_in, ldsP + r * WR_RP, off + urow);\n#endif")
sub("        f32x4 t[4][4];\n#pragma unroll\n        for (int i = 0; i < 4; ++i)", "#pragma unroll\n        for (int i = 0; i < 4; ++i)")
sub("    while (item < a.nblk) {", "    f32x4 t[4][4];\n    bool abl_loaded = false;\n    (void)abl_loaded;\n    while (item < a.nblk) {")
sub("            for (int j = 0; j < 4; ++j) t[i][j] = *(const f32x4*)(ldsP + (j < 2 ? offA : offB) + i * WR_RP + j * 16);",
    "#if WR_ABL != 3 && WR_ABL != 4\n            for (int j = 0; j < 4; ++j) t[i][j] = *(const f32x4*)(ldsP + (j < 2 ? offA : offB) + i * WR_RP + j * 16);\n#else\n"
    "            for (int j = 0; j < 4; ++j)\n                if (!abl_loaded) t[i][j] = *(const f32x4*)(ldsP + (j < 2 ? offA : offB) + i * WR_RP + j * 16);\n#endif")
sub("        const float* wb = wbl + cc * WR_WFL;", "        abl_loaded = true;\n        const float* wb = wbl + cc * WR_WFL;")
sub("#pragma unroll\n                        for (int nb = 0; nb < WR_NB; ++nb) wnx[nb] = *(const f32x4*)(wb + (xi + 1) * (4 * WR_TN * 4) + nb * 64);",
    "#if WR_ABL != 3 && WR_ABL != 4\n#pragma unroll\n                        for (int nb = 0; nb < WR_NB; ++nb) wnx[nb] = *(const f32x4*)(wb + (xi + 1) * (4 * WR_TN * 4) + nb * 64);\n#endif")
sub("            t[0][j] = aesr_sub4(d0, d2);", "#if WR_ABL == 2 || WR_ABL == 4\n            continue;\n#endif\n            t[0][j] = aesr_sub4(d0, d2);")
sub("#define WR_V(i, j) ((j) == 0 ?", "#if WR_ABL == 2 || WR_ABL == 4\n#define WR_V(i, j) (t[i][j])\n#else\n#define WR_V(i, j) ((j) == 0 ?")
sub("aesr_sub4(t[i][2], t[i][1]) : aesr_sub4(t[i][1], t[i][3]))\n", "aesr_sub4(t[i][2], t[i][1]) : aesr_sub4(t[i][1], t[i][3]))\n#endif\n")
sub("        cc = 0;\n        // ---- item finished:",
    "        cc = 0;\n#if WR_ABL == 4 || WR_ABL == 5\n#pragma unroll\n        for (int xi = 0; xi < 16; ++xi)\n#pragma unroll\n"
    "            for (int nb = 0; nb < WR_NB; ++nb) asm volatile(\"\" ::\"v\"(acc[xi][nb]));\n        after_stores = false;\n        continue;\n#endif\n"
    "        // ---- item finished:")
src = os.path.join(VAR, "conv_wino_res_abl.hip")
open(src, "w").write(s)
objs = [o for o in "aesr_api prep conv_igemm conv_wino conv_wino_ring conv_wgrad conv_wgrad_wino conv_small conv_thin bn bn_fused resample elementwise lpips metrics vif "
        "augment lap comm p2p".split()]
for n in range(6):
    o = os.path.join(VAR, "res_abl%d.o" % n)
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-std=c++17", "--offload-arch=gfx950", "-Wall", "-Wno-unused-result", "-ffp-contract=off",
                    "-fno-slp-vectorize", "-I" + CS, "-DWR_ABL=%d" % n, "-c", src, "-o", o], check=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", os.path.join(VAR, "libaesr_resabl%d.so" % n), o] +
                   [os.path.join(CS, "build", x + ".o") for x in objs] + ["-ldl"], check=True)
    print("built WR_ABL=%d" % n)
