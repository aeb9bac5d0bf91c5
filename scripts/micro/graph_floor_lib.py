#!/usr/bin/env python
"""Per-node cost of captured chains of THIS library's kernels (the fast path of the HIP runtime's graph replay replays pre-built AQL packets:
1.6 us per node for torch's tiny kernels, scripts/micro/graph_floor.py): which of our kernels, if any, keep a graph off that path?

    python scripts/micro/graph_floor_lib.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from superresolution_aniso_mri_amd import _hip as hip  # noqa: E402
from scripts.bench_wino import pack_wino  # noqa: E402

L = hip.lib
N = 200
x = torch.zeros(1024, device="cuda")
y = torch.randn(4096, device="cuda")
g = torch.randn(4096, device="cuda")
o = torch.empty(4096, device="cuda")
w = torch.randn(32, 32, 3, 3, device="cuda") * 0.05
b = torch.zeros(32, device="cuda")
uf = pack_wino(w, 32, 32, 0)
xin = torch.randn(1, 8, 8, 32, device="cuda")
xout = torch.empty(1, 8, 8, 32, device="cuda")


def k_torch():
    x.add_(1.0)


def k_act():
    hip.check(L.aesr_act_bwd(hip.ptr(y), hip.ptr(g), hip.ptr(o), 4096, hip.ACT_LRELU, 0.01, hip.stream()), "act_bwd")


def k_conv():
    hip.check(L.aesr_conv2d_wino_fwd(hip.ptr(xin), hip.ptr(uf), hip.ptr(b), hip.ptr(xout), 1, 8, 8, 32, 32, 1, 0.01, hip.stream()), "conv")


def chain_of(*ks):
    def fn():
        for i in range(N):
            ks[i % len(ks)]()
    return fn


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps / N


def graphed(fn):
    fn()
    torch.cuda.synchronize()
    gg = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gg):
        fn()
    return gg.replay


for name, ks in [("torch add_", (k_torch,)), ("aesr_act_bwd (4096 elements)", (k_act,)), ("aesr_conv2d_wino_fwd 1 x 8 x 8 x 32 (resident-filter kernel, 64+ KB of dynamic LDS)", (k_conv,)),
                 ("torch add_ / aesr_act_bwd alternating", (k_torch, k_act)), ("torch add_ x 9 + one resident-filter conv per ten nodes", (k_torch,) * 9 + (k_conv,))]:
    print("%-95s graph replay %.2f us per node" % (name + ":", timed(graphed(chain_of(*ks)))))
