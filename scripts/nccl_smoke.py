#!/usr/bin/env python
"""RCCL sanity on one GPU (world_size 1): the collectives, dtypes and call pattern of parallel.DataParallelContext, eager and
captured in a HIP graph (CAPTURE_MODE=global|thread_local|relaxed).  Run: python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 scripts/nccl_smoke.py"""
import os
import time

import torch
import torch.distributed as dist

torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
dist.init_process_group("nccl")
a = torch.arange(8, dtype=torch.float64, device="cuda")
b = torch.ones(443777, dtype=torch.float32, device="cuda")
dist.all_reduce(a)
dist.all_reduce(b)
dist.broadcast(b, src=0)
dist.barrier()
torch.cuda.synchronize()
print("eager ok", a.sum().item(), b.sum().item())
t0 = time.perf_counter()
for _ in range(100):
    dist.all_reduce(a)
torch.cuda.synchronize()
print("all_reduce fp64[8] host+device us/call: %.1f" % ((time.perf_counter() - t0) * 1e4))
try:
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            dist.all_reduce(a)
    torch.cuda.current_stream().wait_stream(s)
    mode = os.environ.get("CAPTURE_MODE", "global")
    print("capture mode", mode, flush=True)
    with torch.cuda.graph(g, capture_error_mode=mode):
        a.mul_(2.0)
        dist.all_reduce(a)
        b.add_(1.0)
        dist.all_reduce(b)
    g.replay()
    g.replay()
    torch.cuda.synchronize()
    print("graph capture of all_reduce ok", a.sum().item(), b[0].item())
except Exception as e:       # noqa: BLE001
    print("graph capture of all_reduce FAILED: %r" % (e,))
dist.destroy_process_group()
