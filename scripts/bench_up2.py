#!/usr/bin/env python
"""Folded nearest Upsample(x2): time of the three kernels with the half-resolution input against the same kernels on a
materialised upsampled tensor (dec.6 / dec.12 shapes of the C2 step)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superresolution_aniso_mri_amd import _hip as hip  # noqa: E402
from scripts.bench_conv import timeit  # noqa: E402
from scripts.bench_wino import pack_wino  # noqa: E402

L = hip.lib
for name, N, H, W, Cin, Cout in (("dec.6 64->32 @80", 36, 80, 80, 64, 32), ("dec.12 32->32 @160", 36, 160, 160, 32, 32)):
    xh = torch.randn(N, H // 2, W // 2, Cin, device="cuda")
    xu = xh.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2).contiguous()
    dy = torch.randn(N, H, W, Cout, device="cuda")
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
    b = torch.zeros(Cout, device="cuda")
    uf, ub = pack_wino(w, Cout, Cin, 0), pack_wino(w, Cout, Cin, 1)
    out = torch.empty(N, H, W, Cout, device="cuda")
    dx, dxh = torch.empty(N, H, W, Cin, device="cuda"), torch.empty(N, H // 2, W // 2, Cin, device="cuda")
    dw, db = torch.empty_like(w), torch.empty_like(b)
    ws = torch.empty(L.aesr_conv2d_wgrad_workspace_floats(N, H, W, Cin, Cout, 3, 1), device="cuda")
    t = {}
    t["fwd"] = timeit(lambda: hip.check(L.aesr_conv2d_wino_fwd(hip.ptr(xu), hip.ptr(uf), hip.ptr(b), hip.ptr(out), N, H, W, Cin, Cout, 1, 0.01, hip.stream()), "f"))
    t["fwd_up2"] = timeit(lambda: hip.check(L.aesr_conv2d_wino_fwd_up2(hip.ptr(xh), hip.ptr(uf), hip.ptr(b), hip.ptr(out), N, H, W, Cin, Cout, 1, 0.01, hip.stream()), "f"))
    t["dgrad"] = timeit(lambda: hip.check(L.aesr_conv2d_wino_dgrad(hip.ptr(dy), hip.ptr(ub), None, hip.ptr(dx), N, H, W, Cin, Cout, 0, 0.0, hip.stream()), "d"))
    t["dgrad_sum2"] = timeit(lambda: hip.check(L.aesr_conv2d_wino_dgrad_sum2(hip.ptr(dy), hip.ptr(ub), hip.ptr(dxh), N, H, W, Cin, Cout, hip.stream()), "d"))
    t["wgrad"] = timeit(lambda: hip.check(L.aesr_conv2d_wgrad(hip.ptr(xu), hip.ptr(dy), hip.ptr(dw), hip.ptr(db), hip.ptr(ws), N, H, W, Cin, Cout, 3, 1, hip.stream()), "w"))
    t["wgrad_up2"] = timeit(lambda: hip.check(L.aesr_conv2d_wgrad_up2(hip.ptr(xh), hip.ptr(dy), hip.ptr(dw), hip.ptr(db), hip.ptr(ws), N, H, W, Cin, Cout, hip.stream()), "w"))
    print(name, " ".join("%s %.1f us" % (k, v * 1e6) for k, v in t.items()), flush=True)
