#!/usr/bin/env python
"""Time the BatchNorm kernels at the C2 shapes (HBM-bound: GB/s per launch): bench_bn.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superresolution_aniso_mri_amd import _hip as hip  # noqa: E402

L = hip.lib
#        name      N   H    C  mode (0 none, 1 pool, 2 up)
CASES = [("enc.5", 24, 162, 32, 1), ("enc.11", 24, 81, 64, 1), ("dec.4", 36, 40, 64, 2), ("dec.10", 36, 80, 32, 2)]
for name, N, H, C, mode in CASES:
    y = torch.randn(N, H, H, C, device="cuda")
    Ho = H // 2 if mode == 1 else (2 * H if mode == 2 else H)
    g = torch.randn(N, Ho, Ho, C, device="cuda")
    mean, invstd, scale = torch.zeros(1, C, device="cuda"), torch.ones(1, C, device="cuda"), torch.ones(1, C, device="cuda")
    partial = torch.empty(hip.BN_NWG * 2 * C, device="cuda")
    sums = torch.empty((1, 2, C), dtype=torch.float64, device="cuda")
    coef = torch.empty((1, 2, C), device="cuda")
    dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    dpre = torch.empty_like(y)
    ns = hip.int_array([0, N])
    counts = hip.double_array([N * H * H])

    def reduce():
        hip.check(L.aesr_bn_bwd_reduce(hip.ptr(g), hip.ptr(y), hip.ptr(mean), hip.ptr(invstd), hip.ptr(partial), hip.ptr(sums), N, H, H, C, mode, 1, ns,
                                       hip.stream()), "reduce")

    def apply():
        hip.check(L.aesr_bn_bwd_apply(hip.ptr(g), hip.ptr(y), hip.ptr(mean), hip.ptr(invstd), hip.ptr(scale), hip.ptr(sums), counts, hip.ptr(coef),
                                      hip.ptr(dg), hip.ptr(db), hip.ptr(dpre), N, H, H, C, mode, 1, 0.01, 1, ns, hip.stream()), "apply")
    out = torch.empty_like(g)
    shift = torch.zeros(1, C, device="cuda")

    def fwd_apply():
        hip.check(L.aesr_bn_apply(hip.ptr(y), hip.ptr(scale), hip.ptr(shift), hip.ptr(out), N, H, H, C, mode, 1, ns, hip.stream()), "fwd apply")

    def fwd_stats():
        hip.check(L.aesr_bn_stats(hip.ptr(y), hip.ptr(partial), hip.ptr(sums), H * H, C, 1, ns, hip.stream()), "fwd stats")
    yb, gb = y.numel() * 4 / 1e6, g.numel() * 4 / 1e6
    for fn, mb, label in ((fwd_stats, yb, "fwd_stats(+sum kernel)"), (fwd_apply, yb + gb, "fwd_apply"),
                          (reduce, yb + gb, "bwd_reduce(+sum kernel)"), (apply, 2 * yb + gb, "bwd_apply(+finalize)")):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print("%-7s %-24s %7.1f us  %6.0f MB  %5.2f TB/s" % (name, label, us, mb, mb / us))
