#!/usr/bin/env python
"""Time ONE Winograd forward layer (HIP events around 10 back-to-back launches, best of 5): time_one.py fwd N H W Cin Cout -> microseconds."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superresolution_aniso_mri_amd import _hip as hip  # noqa: E402
from scripts.bench_conv import timeit  # noqa: E402
from scripts.bench_wino import pack_wino  # noqa: E402

L = hip.lib
N, H, W, Cin, Cout = [int(v) for v in sys.argv[2:7]]
x = torch.randn(N, H, W, Cin, device="cuda")
w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
b = torch.zeros(Cout, device="cuda")
u = pack_wino(w, Cout, Cin, 0)
out = torch.empty(N, H, W, Cout, device="cuda")
fn = lambda: hip.check(L.aesr_conv2d_wino_fwd(hip.ptr(x), hip.ptr(u), hip.ptr(b), hip.ptr(out), N, H, W, Cin, Cout, 1, 0.01, hip.stream()), "fwd")
print("%.1f" % (timeit(fn) * 1e6))
