#!/usr/bin/env python
"""Stress check of aesr_conv2d_wgrad at BASELINE sizes: run-to-run bitwise equality and the error against fp64.
check_wgrad_big.py N H W Cin Cout [up2] [mode]   (mode: randn | common = dy with a large common-mode part)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superresolution_aniso_mri_amd import _hip as hip  # noqa: E402

L = hip.lib
N, H, W, Cin, Cout = [int(v) for v in sys.argv[1:6]]
up2 = len(sys.argv) > 6 and sys.argv[6] == "1"
mode = sys.argv[7] if len(sys.argv) > 7 else "randn"
g = torch.Generator(device="cuda").manual_seed(7)
xs = (N, H // 2, W // 2, Cin) if up2 else (N, H, W, Cin)
x = torch.randn(xs, device="cuda", generator=g)
dy = torch.randn((N, H, W, Cout), device="cuda", generator=g)
if mode == "common":
    x = torch.nn.functional.leaky_relu(x + 0.5, 0.01)
    dy = 1.0 + 1e-2 * dy
ws = torch.empty(L.aesr_conv2d_wgrad_workspace_floats(N, H, W, Cin, Cout, 3, 1), device="cuda")
outs = []
for it in range(4):
    dw = torch.full((Cout, Cin, 3, 3), float("nan"), device="cuda")
    db = torch.full((Cout,), float("nan"), device="cuda")
    ws.fill_(float("nan"))
    if up2:
        hip.check(L.aesr_conv2d_wgrad_up2(hip.ptr(x), hip.ptr(dy), hip.ptr(dw), hip.ptr(db), hip.ptr(ws), N, H, W, Cin, Cout, hip.stream()), "wgrad_up2")
    else:
        hip.check(L.aesr_conv2d_wgrad(hip.ptr(x), hip.ptr(dy), hip.ptr(dw), hip.ptr(db), hip.ptr(ws), N, H, W, Cin, Cout, 3, 1, hip.stream()), "wgrad")
    torch.cuda.synchronize()
    outs.append((dw, db))
same = all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs[1:])
xn = x.permute(0, 3, 1, 2).double()
if up2:
    xn = torch.nn.functional.interpolate(xn, scale_factor=2, mode="nearest")
ref_w = torch.nn.grad.conv2d_weight(xn, (Cout, Cin, 3, 3), dy.permute(0, 3, 1, 2).double(), padding=1)
ref_b = dy.double().sum((0, 1, 2))
ew = float((outs[0][0].double() - ref_w).norm() / ref_w.norm())
eb = float((outs[0][1].double() - ref_b).norm() / ref_b.norm())
print("N=%d %dx%d %d->%d up2=%d %s: repeatable=%s  dW rel err %.2e  db rel err %.2e  |dW| %.4g  max run-to-run diff %.3g" % (
    N, H, W, Cin, Cout, up2, mode, same, ew, eb, float(ref_w.norm()), max(float((outs[0][0] - o[0]).abs().max()) for o in outs[1:])))
