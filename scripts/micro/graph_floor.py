#!/usr/bin/env python
"""Per-node cost of a captured chain of tiny dependent kernels against the same chain launched eagerly, under the HIP runtime's graph switches
(run once per setting of DEBUG_CLR_GRAPH_PACKET_CAPTURE / DEBUG_HIP_GRAPH_BATCH_SIZE: the runtime reads them at start-up).

    python scripts/micro/graph_floor.py [nodes]"""
import os
import sys
import time

import torch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
x = torch.zeros(1024, device="cuda")


def chain():
    for _ in range(N):
        x.add_(1.0)           # one tiny dependent kernel per call


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps / N, (time.perf_counter() - t0) * 1e6 / reps / N


chain()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    chain()
dev_e, wall_e = timed(chain)
dev_g, wall_g = timed(g.replay)
print("PACKET_CAPTURE=%s BATCH_SIZE=%s: %d dependent tiny kernels: eager %.2f us per kernel on the device clock (%.2f wall), graph replay %.2f us per node (%.2f wall)" % (
    os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "-"), os.environ.get("DEBUG_HIP_GRAPH_BATCH_SIZE", "-"), N, dev_e, wall_e, dev_g, wall_g))

# what a tiny kernel costs BEHIND a big writer (its predecessor's dirty lines leave the L2s at the kernel boundary): chain of (big, tiny) pairs against the bigs alone
big = torch.zeros(36 * 160 * 160 * 32, device="cuda")          # 118 MB, the largest activation of BASELINE configuration 2
M = 40


def bigs():
    for _ in range(M):
        big.add_(1.0)


def pairs():
    for _ in range(M):
        big.add_(1.0)
        x.add_(1.0)


def graphed(fn):
    fn()
    torch.cuda.synchronize()
    gg = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gg):
        fn()
    return gg.replay


N = M
tb, _ = timed(graphed(bigs))
tp, _ = timed(graphed(pairs))
print("graph of %d x (118 MB in-place add): %.2f us each; with a tiny kernel behind each: %.2f us per pair -> the tiny kernel costs %.2f us there" % (M, tb, tp, tp - tb))

# the same with TEN DIFFERENT tiny kernels in rotation behind the big one (cold code / arguments each time?)
ops = [lambda: x.add_(1.0), lambda: x.mul_(1.0), lambda: x.sin_(), lambda: x.cos_(), lambda: x.abs_(), lambda: x.neg_(), lambda: x.exp_(), lambda: x.tanh_(),
       lambda: x.sigmoid_(), lambda: x.sqrt_()]


def pairs_distinct():
    for i in range(M):
        big.add_(1.0)
        ops[i % 10]()


def tiny_distinct():
    for i in range(N):
        ops[i % 10]()


td, _ = timed(graphed(pairs_distinct))
print("with ten DIFFERENT tiny kernels in rotation behind the big one: %.2f us per pair -> %.2f us per tiny kernel" % (td, td - tb))
N = 200
tt, _ = timed(graphed(tiny_distinct))
print("chain of 200 tiny kernels, ten different ones in rotation: %.2f us per node" % tt)
