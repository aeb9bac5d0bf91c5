#!/usr/bin/env python
"""HBM-cold timing of the two expand-type thin kernels at the shapes of the C2 step and of the inference leg (see r04_thin_eth.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superresolution_aniso_mri_amd import _hip as hip  # noqa: E402

L = hip.lib
scrub = torch.zeros(150 * 1000 * 1000, device="cuda")


def cold(fn, reps=7):
    ts = []
    for _ in range(reps):
        scrub.add_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


wst, bst = torch.randn(32, device="cuda"), torch.randn(32, device="cuda")
w1, b1 = torch.randn(32, 32, 3, 3, device="cuda") * 0.05, torch.zeros(32, device="cuda")
folded = torch.empty(L.aesr_stemconv_folded_floats(32), device="cuda")
hip.check(L.aesr_stemconv_fold(hip.ptr(wst), hip.ptr(bst), hip.ptr(w1), hip.ptr(folded), 32, 32, hip.stream()), "f")
for n, h in ((36, 160), (30, 224), (6, 160)):
    img = torch.rand(n, h, h, device="cuda")
    so = torch.empty(n, h + 2, h + 2, 32, device="cuda")
    t = cold(lambda: hip.check(L.aesr_stemconv_fwd(hip.ptr(img), hip.ptr(folded), hip.ptr(b1), hip.ptr(so), n, h, h, 32, 1, 1, 0.01, hip.stream()), "s"))
    print("  stemconv_fwd %3d x %d^2 -> %d^2 x 32   %7.1f us  %5.2f TB/s written" % (n, h, h + 2, t, so.numel() * 4 / t / 1e6))
for n, h in ((24, 160), (4, 160)):
    x = torch.randn(n, h, h, 32, device="cuda")
    w = torch.randn(1, 32, 3, 3, device="cuda") * 0.1
    dy = torch.randn(n, h, h, 1, device="cuda")
    dx = torch.empty_like(x)
    wsf = torch.empty(9 * 32, device="cuda")
    t = cold(lambda: hip.check(L.aesr_conv2d_cout1_dgrad(hip.ptr(dy), hip.ptr(w), hip.ptr(x), hip.ptr(dx), hip.ptr(wsf), n, h, h, 32, 1, 0.01, hip.stream()), "d"))
    print("  cout1_dgrad  %3d x %d^2 x 32 (mask read + write)   %7.1f us  %5.2f TB/s" % (n, h, t, 2 * x.numel() * 4 / t / 1e6))
