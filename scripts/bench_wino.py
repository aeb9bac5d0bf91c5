#!/usr/bin/env python
"""Winograd F(2x2,3x3) kernel (csrc/conv_wino.hip) at the layer shapes of the C2/C3 step: result against the implicit-GEMM
kernel and (small batch) against an fp64 torch convolution, then time per layer next to conv_igemm_f32.
    python scripts/bench_wino.py [ae|vgg|all] [--check-only]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superresolution_aniso_mri_amd import _hip as hip  # noqa: E402
from scripts.bench_conv import AE, VGG, PEAK, timeit  # noqa: E402

L = hip.lib


def pack_wino(w, Cout, Cin, transpose):
    buf = torch.empty(L.aesr_conv2d_wino_packed_floats(Cout, Cin, transpose), device="cuda")
    job = (hip.PackJob * 1)(hip.PackJob(w.data_ptr(), buf.data_ptr(), Cout, Cin, 3, transpose))
    hip.check(L.aesr_conv2d_wino_pack_many(job, 1, hip.stream()), "wino_pack")
    return buf


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def main():
    which = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "ae"
    layers = {"ae": AE, "vgg": VGG, "all": AE + VGG}[which]
    tot = {"fwd": [0, 0, 0], "dgrad": [0, 0, 0]}
    for name, N, H, W, Cin, Cout in layers:
        g = torch.Generator(device="cuda").manual_seed(H + Cin)
        x = torch.randn(N, H, W, Cin, device="cuda", generator=g)
        dy = torch.randn(N, H, W, Cout, device="cuda", generator=g)
        w = torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) * 0.05
        b = torch.randn(Cout, device="cuda", generator=g)
        pf = torch.empty(L.aesr_conv2d_packed_floats(Cout, Cin, 3, 0), device="cuda")
        pb = torch.empty(L.aesr_conv2d_packed_floats(Cout, Cin, 3, 1), device="cuda")
        hip.check(L.aesr_conv2d_pack(hip.ptr(w), hip.ptr(pf), Cout, Cin, 3, 0, hip.stream()), "pack")
        hip.check(L.aesr_conv2d_pack(hip.ptr(w), hip.ptr(pb), Cout, Cin, 3, 1, hip.stream()), "pack")
        uf, ub = pack_wino(w, Cout, Cin, 0), pack_wino(w, Cout, Cin, 1)
        out_i, out_w = torch.empty(N, H, W, Cout, device="cuda"), torch.full((N, H, W, Cout), float("nan"), device="cuda")
        dx_i, dx_w = torch.empty(N, H, W, Cin, device="cuda"), torch.full((N, H, W, Cin), float("nan"), device="cuda")
        f_i = lambda: hip.check(L.aesr_conv2d_fwd(hip.ptr(x), hip.ptr(pf), hip.ptr(b), hip.ptr(out_i), N, H, W, Cin, Cout, 3, 1, 1, 0.01, hip.stream()), "fwd")
        # as the product calls it: with the workspace the library asks for (channel split of small / deep layers)
        nwf, nwd = L.aesr_conv2d_wino_workspace_floats(N, H, W, Cin, Cout, 0), L.aesr_conv2d_wino_workspace_floats(N, H, W, Cin, Cout, 1)
        wsf = torch.empty(nwf, device="cuda") if nwf else None
        wsd = torch.empty(nwd, device="cuda") if nwd else None
        f_w = lambda: hip.check(L.aesr_conv2d_wino_fwd_ws(hip.ptr(x), hip.ptr(uf), hip.ptr(b), hip.ptr(out_w), hip.ptr(wsf), nwf, N, H, W, Cin, Cout, 1, 0.01, hip.stream()), "wino_fwd")
        d_i = lambda: hip.check(L.aesr_conv2d_dgrad(hip.ptr(dy), hip.ptr(pb), hip.ptr(x), hip.ptr(dx_i), N, H, W, Cin, Cout, 3, 1, 1, 0.01, hip.stream()), "dgrad")
        d_w = lambda: hip.check(L.aesr_conv2d_wino_dgrad_ws(hip.ptr(dy), hip.ptr(ub), hip.ptr(x), hip.ptr(dx_w), hip.ptr(wsd), nwd, N, H, W, Cin, Cout, 1, 0.01, hip.stream()), "wino_dgrad")
        f_i(); f_w(); d_i(); d_w()
        torch.cuda.synchronize()
        # fp64 reference on two images
        xr = x[:2].permute(0, 3, 1, 2).double()
        ref = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(xr, w.double(), b.double(), padding=1), 0.01).permute(0, 2, 3, 1)
        dref = torch.nn.grad.conv2d_input(xr.shape, w.double(), dy[:2].permute(0, 3, 1, 2).double(), padding=1).permute(0, 2, 3, 1)
        dref = dref * torch.where(x[:2].double() > 0, 1.0, 0.01)
        line = "%-22s fwd: wino-vs-igemm %.1e | vs fp64: wino %.1e igemm %.1e || dgrad: wino-vs-igemm %.1e | vs fp64: wino %.1e igemm %.1e" % (
            name, rel(out_w, out_i), rel(out_w[:2], ref), rel(out_i[:2], ref), rel(dx_w, dx_i), rel(dx_w[:2], dref), rel(dx_i[:2], dref))
        print(line, flush=True)
        if "--check-only" in sys.argv:
            continue
        fl = 2.0 * N * H * W * Cin * Cout * 9
        ti_f, tw_f, ti_d, tw_d = timeit(f_i), timeit(f_w), timeit(d_i), timeit(d_w)
        for k, ti, tw in (("fwd", ti_f, tw_f), ("dgrad", ti_d, tw_d)):
            tot[k][0] += fl
            tot[k][1] += ti
            tot[k][2] += tw
        print("    %6.2f GF | fwd igemm %7.1f us (%4.1f%%) wino %7.1f us (%5.1f%% eff, x%.2f) | dgrad igemm %7.1f us (%4.1f%%) wino %7.1f us (%5.1f%% eff, x%.2f)" % (
            fl / 1e9, ti_f * 1e6, 100 * fl / ti_f / 1e12 / PEAK, tw_f * 1e6, 100 * fl / tw_f / 1e12 / PEAK, ti_f / tw_f,
            ti_d * 1e6, 100 * fl / ti_d / 1e12 / PEAK, tw_d * 1e6, 100 * fl / tw_d / 1e12 / PEAK, ti_d / tw_d), flush=True)
    for k, (f, ti, tw) in tot.items():
        if ti:
            print("TOTAL %-6s %7.1f GF: igemm %8.1f us = %5.1f TF (%4.1f%%) | wino %8.1f us = %5.1f TF effective (%5.1f%% of fp32 MFMA peak), x%.2f" % (
                k, f / 1e9, ti * 1e6, f / ti / 1e12, 100 * f / ti / 1e12 / PEAK, tw * 1e6, f / tw / 1e12, 100 * f / tw / 1e12 / PEAK, ti / tw))


if __name__ == "__main__":
    main()
