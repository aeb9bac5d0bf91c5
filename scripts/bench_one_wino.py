#!/usr/bin/env python
"""Run ONE Winograd convolution layer a few times (for rocprofv3 --pmc runs): bench_one_wino.py fwd|dgrad|wgrad N H W Cin Cout [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superresolution_aniso_mri_amd import _hip as hip  # noqa: E402
from scripts.bench_wino import pack_wino  # noqa: E402

L = hip.lib
kind = sys.argv[1]
N, H, W, Cin, Cout = [int(v) for v in sys.argv[2:7]]
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 5
x = torch.randn(N, H, W, Cin, device="cuda")
dy = torch.randn(N, H, W, Cout, device="cuda")
w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
b = torch.zeros(Cout, device="cuda")
uf, ub = pack_wino(w, Cout, Cin, 0), pack_wino(w, Cout, Cin, 1)
out = torch.empty(N, H, W, Cout, device="cuda")
dx = torch.empty(N, H, W, Cin, device="cuda")
dw, db = torch.empty_like(w), torch.empty_like(b)
ws = torch.empty(L.aesr_conv2d_wgrad_workspace_floats(N, H, W, Cin, Cout, 3, 1), device="cuda")
for _ in range(iters):
    if kind == "fwd":
        hip.check(L.aesr_conv2d_wino_fwd(hip.ptr(x), hip.ptr(uf), hip.ptr(b), hip.ptr(out), N, H, W, Cin, Cout, 1, 0.01, hip.stream()), "fwd")
    elif kind == "dgrad":
        hip.check(L.aesr_conv2d_wino_dgrad(hip.ptr(dy), hip.ptr(ub), hip.ptr(x), hip.ptr(dx), N, H, W, Cin, Cout, 1, 0.01, hip.stream()), "dgrad")
    else:
        hip.check(L.aesr_conv2d_wgrad(hip.ptr(x), hip.ptr(dy), hip.ptr(dw), hip.ptr(db), hip.ptr(ws), N, H, W, Cin, Cout, 3, 1, hip.stream()), "wgrad")
torch.cuda.synchronize()
print("done")
