#!/usr/bin/env python
"""Per-launch budget of the MFMA convolution kernels at the shard of a data-parallel rank (2 triplets: 6 images forward, 4 backward;
round-3 verdict, next 1a): for every 3x3 layer of the ACDC auto-encoder the kernel the planner picks, its time (HIP events around
back-to-back launches), the MFMA work it executes, and the time the matrix cores need for that work if it were spread evenly over the
1024 SIMDs ("floor") and as it is actually dealt out -- items per busiest SIMD x MFMAs per item x 32 cycles at 2.1 GHz ("dealt").
   shard_budget.py [N_fwd N_bwd]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superresolution_aniso_mri_amd import _hip as hip  # noqa: E402
from scripts.bench_conv import timeit  # noqa: E402
from scripts.bench_wino import pack_wino  # noqa: E402

L = hip.lib
NF = int(sys.argv[1]) if len(sys.argv) > 1 else 6
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 4
KIND = {1: "conv_wino_f32", 2: "conv_wino_res_f32", 3: "conv_wino_ring_f32"}
LAYERS = [("enc.3 32->32 @162", 162, 32, 32, NF, NB), ("enc.7 32->64 @81", 81, 32, 64, NF, NB), ("enc.9 64->64 @81", 81, 64, 64, NF, NB),
          ("enc.13 64->128 @40", 40, 64, 128, NF, NB), ("enc.15 128->128 @40", 40, 128, 128, NF, NB), ("dec.0 128->64 @40", 40, 128, 64, NF, NF),
          ("dec.2 64->64 @40", 40, 64, 64, NF, NF), ("dec.6 64->32 @80 (up2)", 80, 64, 32, NF, NF), ("dec.8 32->32 @80", 80, 32, 32, NF, NF),
          ("dec.12 32->32 @160 (up2)", 160, 32, 32, NF, NF)]
GHZ = 2.1


def floor_us(n, h, cin, cout):
    """MFMA time if the layer's Winograd work were spread evenly over 1024 SIMDs: tiles x 16 positions x (cin/4) k-steps x (cout/16)
    MFMAs of 32 cycles (v_mfma_f32_16x16x4_f32: 16 tiles x 16 couts x 4 cin per instruction)."""
    tiles = n * ((h + 1) // 2) ** 2
    mfmas = (tiles / 16.0) * 16 * (cin / 4.0) * (cout / 16.0)
    return mfmas * 32 / 1024 / (GHZ * 1e3), mfmas


print("Winograd F(2x2,3x3) layers of the ACDC auto-encoder at %d images forward / %d backward (MI355X, events around 10 back-to-back launches)" % (NF, NB))
print("%-26s %-9s %-20s %8s %8s %8s %7s" % ("layer", "dir", "kernel", "time us", "floor us", "GF exec", "TF exec"))
tot = {}
for name, h, cin, cout, nf, nb in LAYERS:
    for direction, n in (("fwd", nf), ("dgrad", nb)):
        tr = 1 if direction == "dgrad" else 0
        kin, nout = (cout, cin) if tr else (cin, cout)
        x = torch.randn(n, h, h, kin, device="cuda")
        w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
        u = pack_wino(w, cout, cin, tr)
        out = torch.empty(n, h, h, nout, device="cuda")
        nws = L.aesr_conv2d_wino_workspace_floats(n, h, h, cin, cout, tr)
        ws = torch.empty(nws, device="cuda") if nws else None
        bias = torch.zeros(nout, device="cuda")
        if tr:
            ys = torch.randn(n, h, h, nout, device="cuda")
            fn = lambda: hip.check(L.aesr_conv2d_wino_dgrad_ws(hip.ptr(x), hip.ptr(u), hip.ptr(ys), hip.ptr(out), hip.ptr(ws), nws, n, h, h, cin, cout, 1, 0.01, hip.stream()), "d")
        else:
            fn = lambda: hip.check(L.aesr_conv2d_wino_fwd_ws(hip.ptr(x), hip.ptr(u), hip.ptr(bias), hip.ptr(out), hip.ptr(ws), nws, n, h, h, cin, cout, 1, 0.01, hip.stream()), "f")
        t = timeit(fn) * 1e6
        fl, mf = floor_us(n, h, kin, nout)
        kind = KIND.get(L.aesr_conv2d_wino_kernel(n, h, h, cin, cout, 3, 1, tr), "?") + (" +split" if nws else "")
        gf = mf * 2 * 16 * 16 * 4 / 1e9
        print("%-26s %-9s %-20s %8.1f %8.1f %8.3f %7.1f" % (name, direction, kind, t, fl, gf, gf / t * 1e3))
        d = tot.setdefault(direction, [0.0, 0.0, 0.0])
        d[0] += t; d[1] += fl; d[2] += gf
    # weight gradient (Winograd form where both channel counts are multiples of 32)
    n = nb
    x = torch.randn(n, h, h, cin, device="cuda")
    dy = torch.randn(n, h, h, cout, device="cuda")
    wsz = L.aesr_conv2d_wgrad_workspace_floats(n, h, h, cin, cout, 3, 1)
    ws = torch.empty(wsz, device="cuda")
    fn = lambda: hip.check(L.aesr_conv2d_wgrad_partial(hip.ptr(x), hip.ptr(dy), hip.ptr(ws), n, h, h, cin, cout, 3, 1, 0, hip.stream()), "w")
    t = timeit(fn) * 1e6
    fl, mf = floor_us(n, h, cin, cout)
    gf = mf * 2 * 16 * 16 * 4 / 1e9
    print("%-26s %-9s %-20s %8.1f %8.1f %8.3f %7.1f   (slabs %.1f MB)" % (name, "wgrad", "conv_wgrad_wino_f32", t, fl, gf, gf / t * 1e3, wsz * 4 / 1e6))
    d = tot.setdefault("wgrad", [0.0, 0.0, 0.0])
    d[0] += t; d[1] += fl; d[2] += gf
for k, (t, fl, gf) in tot.items():
    print("TOTAL %-6s %8.1f us measured, %7.1f us at the MFMA rate evenly spread (%.0f %%), %.2f GF executed = %.1f TF" % (k, t, fl, 100 * fl / t, gf, gf / t * 1e3))
