#!/usr/bin/env python
"""Channel split of the ring kernel at small shards: per layer (forward; data gradient with --dgrad), replayed-graph time of
the first streamed kernel, the ring kernel unsplit and with forced splits 2..16, and what the launcher plans by itself.
    AESR_BENCH_N=6 python scripts/bench_ksplit.py [vgg|ae|all] [--dgrad]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superresolution_aniso_mri_amd import _hip as hip  # noqa: E402
from scripts.bench_conv import AE, VGG  # noqa: E402
from scripts.bench_wino import pack_wino  # noqa: E402

L = hip.lib
REPS = 20


def graph_time(fn):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REPS):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5 / REPS * 1e3        # us per launch (+ reduce)


def main():
    which = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "vgg"
    dgrad = "--dgrad" in sys.argv
    layers = [l for l in {"ae": AE, "vgg": VGG, "all": AE + VGG}[which] if l[4] >= 64 or l[5] >= 64]
    print("%-22s %4s | %8s %8s | %s | %8s  plan" % ("layer", "N", "first", "ring S=1", " ".join("%8s" % ("S=%d" % s) for s in (2, 4, 8, 16)), "planned"))
    tot = {}
    for name, N, H, W, Cin, Cout in layers:
        x = torch.randn(N, H, W, Cout if dgrad else Cin, device="cuda")
        w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
        b = torch.randn(Cout, device="cuda")
        up = pack_wino(w, Cout, Cin, 1 if dgrad else 0)
        out = torch.empty(N, H, W, Cin if dgrad else Cout, device="cuda")
        mask = torch.randn_like(out)
        kin = Cout if dgrad else Cin
        if L.aesr_conv2d_wino_kernel(N, H, W, Cin, Cout, 3, 1, 1 if dgrad else 0) == 2:
            continue                                     # resident-filter kernel: not a streamed layer

        def run(ws, nws):
            if dgrad:
                hip.check(L.aesr_conv2d_wino_dgrad_ws(hip.ptr(x), hip.ptr(up), hip.ptr(mask), hip.ptr(out), hip.ptr(ws), nws, N, H, W, Cin, Cout, 2, 0.0,
                                                      hip.stream()), "dgrad_ws")
            else:
                hip.check(L.aesr_conv2d_wino_fwd_ws(hip.ptr(x), hip.ptr(up), hip.ptr(b), hip.ptr(out), hip.ptr(ws), nws, N, H, W, Cin, Cout, 2, 0.0,
                                                    hip.stream()), "fwd_ws")

        res = []
        os.environ.pop("AESR_RING_KSPLIT", None)
        os.environ["AESR_WINO_RING"] = "0"
        res.append(graph_time(lambda: run(None, 0)))
        os.environ["AESR_WINO_RING"] = "2"
        res.append(graph_time(lambda: run(None, 0)))
        for s in (2, 4, 8, 16):
            os.environ["AESR_RING_KSPLIT"] = str(s)
            nws = L.aesr_conv2d_wino_workspace_floats(N, H, W, Cin, Cout, 1 if dgrad else 0)
            if nws // out.numel() != s:
                res.append(float("nan"))                 # not a valid split of this layer's chunks
                continue
            ws = torch.empty(nws, device="cuda")
            res.append(graph_time(lambda: run(ws, nws)))
        os.environ.pop("AESR_RING_KSPLIT", None)
        os.environ.pop("AESR_WINO_RING", None)
        nws = L.aesr_conv2d_wino_workspace_floats(N, H, W, Cin, Cout, 1 if dgrad else 0)
        ws = torch.empty(nws, device="cuda") if nws else None
        kid = L.aesr_conv2d_wino_kernel(N, H, W, Cin, Cout, 3, 1, 1 if dgrad else 0)
        res.append(graph_time(lambda: run(ws, nws)))
        best = min(v for v in res[:-1] if v == v)
        print("%-22s %4d | %8.1f %8.1f | %s | %8.1f  kernel %d split %d  (best %.1f)" % (
            name, N, res[0], res[1], " ".join("%8.1f" % v for v in res[2:6]), res[6], kid, nws // out.numel() if nws else 1, best), flush=True)
        for k, v in zip(("first", "ring1", "planned", "best"), (res[0], res[1], res[6], best)):
            tot[k] = tot.get(k, 0.0) + v
    print("TOTAL us: " + "  ".join("%s %.1f" % kv for kv in tot.items()))


if __name__ == "__main__":
    main()
