#!/usr/bin/env python
"""Forward time of the VGG conv4/5 shapes through aesr_conv2d_fwd_ws (AESR_IGEMM_KSPLIT forces the split factor)."""
import sys
import os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superresolution_aniso_mri_amd import _hip as hip  # noqa: E402
L = hip.lib


def t(N, H, Cin, Cout):
    x = torch.randn(N, H, H, Cin, device="cuda")
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.02
    b = torch.zeros(Cout, device="cuda")
    pf = torch.empty(L.aesr_conv2d_packed_floats(Cout, Cin, 3, 0), device="cuda")
    hip.check(L.aesr_conv2d_pack(hip.ptr(w), hip.ptr(pf), Cout, Cin, 3, 0, hip.stream()), "p")
    out = torch.empty(N, H, H, Cout, device="cuda")
    nws = L.aesr_conv2d_workspace_floats(N, H, H, Cin, Cout, 3, 1)
    ws = torch.empty(max(nws, 1), device="cuda")
    f = lambda: hip.check(L.aesr_conv2d_fwd_ws(hip.ptr(x), hip.ptr(pf), hip.ptr(b), hip.ptr(out), hip.ptr(ws) if nws else None, N, H, H, Cin, Cout, 3, 1, 2, 0.0, hip.stream()), "f")
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 100, nws > 0


for shp in [(24, 20, 256, 512), (24, 20, 512, 512), (24, 10, 512, 512), (12, 20, 512, 256), (12, 20, 512, 512), (12, 10, 512, 512)]:
    us, split = t(*shp)
    print(shp, "%.1f us" % us, "split" if split else "")
