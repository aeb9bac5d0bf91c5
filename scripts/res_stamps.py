#!/usr/bin/env python
"""Where a launch of the resident-filter kernel spends its time when every wave has at most one or two work items (round-3 verdict, next 1a:
"put the phase stamps on the 6-image layers and commit the per-launch budget"): for the forward layers of the ACDC auto-encoder that this
kernel serves, at N images -- the time of back-to-back launches (HIP events) and, from ONE stamped launch (AESR_WINO_RES_DBG=1: wall-clock
stamps of every wave's first item, csrc/conv_wino_res.hip), the span of the kernel on the device and the phases of a busy wave.
   res_stamps.py [N ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superresolution_aniso_mri_amd import _hip as hip  # noqa: E402
from scripts.bench_conv import timeit  # noqa: E402
from scripts.bench_wino import pack_wino  # noqa: E402

L = hip.lib
LAYERS = [("enc.3 32->32 @162", 162, 32, 32, 0), ("enc.7 32->64 @81", 81, 32, 64, 0), ("enc.13 64->128 @40", 40, 64, 128, 0), ("dec.2 64->64 @40", 40, 64, 64, 0),
          ("dec.6 64->32 @80 (up2)", 80, 64, 32, 1), ("dec.8 32->32 @80", 80, 32, 32, 0), ("dec.12 32->32 @160 (up2)", 160, 32, 32, 1)]
for n in [int(v) for v in sys.argv[1:]] or [1, 6]:
    for name, h, cin, cout, up2 in LAYERS:
        if L.aesr_conv2d_wino_kernel(n, h, h, cin, cout, 3, 1, 0) != 2:
            continue
        x = torch.randn(n, h // 2, h // 2, cin, device="cuda") if up2 else torch.randn(n, h, h, cin, device="cuda")
        w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
        u = pack_wino(w, cout, cin, 0)
        b = torch.zeros(cout, device="cuda")
        out = torch.empty(n, h, h, cout, device="cuda")
        if up2:
            fn = lambda: hip.check(L.aesr_conv2d_wino_fwd_up2(hip.ptr(x), hip.ptr(u), hip.ptr(b), hip.ptr(out), n, h, h, cin, cout, 1, 0.01, hip.stream()), "f")
        else:
            fn = lambda: hip.check(L.aesr_conv2d_wino_fwd(hip.ptr(x), hip.ptr(u), hip.ptr(b), hip.ptr(out), n, h, h, cin, cout, 1, 0.01, hip.stream()), "f")
        t = timeit(fn) * 1e6
        sys.stderr.flush()
        print("%-26s N=%-2d  %6.1f us per launch (events around back-to-back launches)" % (name, n, t), file=sys.stderr, flush=True)
        os.environ["AESR_WINO_RES_DBG"] = "1"
        fn()
        torch.cuda.synchronize()
        del os.environ["AESR_WINO_RES_DBG"]
