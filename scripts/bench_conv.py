#!/usr/bin/env python
"""Per-layer micro-benchmark of the MFMA convolution kernels (fwd / dgrad / wgrad) at the shapes of the C2/C3 step.
Prints TFLOP/s per layer and the fraction of the fp32-MFMA peak; used to steer kernel tuning."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superresolution_aniso_mri_amd import _hip as hip  # noqa: E402

L = hip.lib
PEAK = 157.3

AE = [  # name, N, H, W, Cin, Cout
    ("enc.1/3 32->32 @162", 36, 162, 162, 32, 32),
    ("enc.7 32->64 @81", 36, 81, 81, 32, 64),
    ("enc.9 64->64 @81", 36, 81, 81, 64, 64),
    ("enc.13 64->128 @40", 36, 40, 40, 64, 128),
    ("enc.15 128->128 @40", 36, 40, 40, 128, 128),
    ("dec.0 128->64 @40", 36, 40, 40, 128, 64),
    ("dec.2 64->64 @40", 36, 40, 40, 64, 64),
    ("dec.6 64->32 @80", 36, 80, 80, 64, 32),
    ("dec.8 32->32 @80", 36, 80, 80, 32, 32),
    ("dec.12 32->32 @160", 36, 160, 160, 32, 32),
]
VGG = [
    ("vgg1_2 64->64 @160", 24, 160, 160, 64, 64),
    ("vgg2_1 64->128 @80", 24, 80, 80, 64, 128),
    ("vgg2_2 128->128 @80", 24, 80, 80, 128, 128),
    ("vgg3_1 128->256 @40", 24, 40, 40, 128, 256),
    ("vgg3_2 256->256 @40", 24, 40, 40, 256, 256),
    ("vgg4_1 256->512 @20", 24, 20, 20, 256, 512),
    ("vgg4_2 512->512 @20", 24, 20, 20, 512, 512),
    ("vgg5_x 512->512 @10", 24, 10, 10, 512, 512),
]


if os.environ.get("AESR_BENCH_N"):          # small-shard experiments: the same layers at another image count
    _n = int(os.environ["AESR_BENCH_N"])
    AE = [(a, _n, c, d, e, f) for a, _, c, d, e, f in AE]
    VGG = [(a, max(1, _n * 2 // 3), c, d, e, f) for a, _, c, d, e, f in VGG]


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "ae"
    layers = {"ae": AE, "vgg": VGG, "all": AE + VGG}[which]
    tot = {"fwd": [0, 0], "dgrad": [0, 0], "wgrad": [0, 0]}
    for name, N, H, W, Cin, Cout in layers:
        x = torch.randn(N, H, W, Cin, device="cuda")
        dy = torch.randn(N, H, W, Cout, device="cuda")
        w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
        b = torch.zeros(Cout, device="cuda")
        pf = torch.empty(L.aesr_conv2d_packed_floats(Cout, Cin, 3, 0), device="cuda")
        pb = torch.empty(L.aesr_conv2d_packed_floats(Cout, Cin, 3, 1), device="cuda")
        hip.check(L.aesr_conv2d_pack(hip.ptr(w), hip.ptr(pf), Cout, Cin, 3, 0, hip.stream()), "pack")
        hip.check(L.aesr_conv2d_pack(hip.ptr(w), hip.ptr(pb), Cout, Cin, 3, 1, hip.stream()), "pack")
        out = torch.empty(N, H, W, Cout, device="cuda")
        dx = torch.empty(N, H, W, Cin, device="cuda")
        dw, db = torch.empty_like(w), torch.empty_like(b)
        ws = torch.empty(L.aesr_conv2d_wgrad_workspace_floats(N, H, W, Cin, Cout, 3, 1), device="cuda")
        fl = 2.0 * N * H * W * Cin * Cout * 9
        t_f = timeit(lambda: hip.check(L.aesr_conv2d_fwd(hip.ptr(x), hip.ptr(pf), hip.ptr(b), hip.ptr(out), N, H, W, Cin, Cout, 3, 1, 1,
                                                         0.01, hip.stream()), "fwd"))
        t_d = timeit(lambda: hip.check(L.aesr_conv2d_dgrad(hip.ptr(dy), hip.ptr(pb), hip.ptr(x), hip.ptr(dx), N, H, W, Cin, Cout, 3, 1, 1,
                                                           0.01, hip.stream()), "dgrad"))
        t_w = timeit(lambda: hip.check(L.aesr_conv2d_wgrad(hip.ptr(x), hip.ptr(dy), hip.ptr(dw), hip.ptr(db), hip.ptr(ws), N, H, W, Cin,
                                                           Cout, 3, 1, hip.stream()), "wgrad"))
        for k, t in (("fwd", t_f), ("dgrad", t_d), ("wgrad", t_w)):
            tot[k][0] += fl
            tot[k][1] += t
        print("%-24s %6.2f GF | fwd %7.1f us %5.1f TF (%4.1f%%) | dgrad %7.1f us %5.1f TF (%4.1f%%) | wgrad %7.1f us %5.1f TF (%4.1f%%)" % (
            name, fl / 1e9, t_f * 1e6, fl / t_f / 1e12, 100 * fl / t_f / 1e12 / PEAK, t_d * 1e6, fl / t_d / 1e12,
            100 * fl / t_d / 1e12 / PEAK, t_w * 1e6, fl / t_w / 1e12, 100 * fl / t_w / 1e12 / PEAK), flush=True)
    for k, (f, t) in tot.items():
        print("TOTAL %-6s %7.1f GF in %8.1f us = %5.1f TF (%4.1f%% of fp32 MFMA peak)" % (k, f / 1e9, t * 1e6, f / t / 1e12,
                                                                                       100 * f / t / 1e12 / PEAK))


if __name__ == "__main__":
    main()
