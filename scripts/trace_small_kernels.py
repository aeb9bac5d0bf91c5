#!/usr/bin/env python
"""Which Python line launches the small torch kernels (fill / copy / cat / elementwise) inside one eager training step:
torch.profiler with stacks, printed per kernel name.  trace_small_kernels.py [c2|c3]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from superresolution_aniso_mri_amd.data_synth import synthetic_batch  # noqa: E402
from superresolution_aniso_mri_amd.kwatsch.get_trainer import get_trainer_dynamic  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
torch.manual_seed(1)
tr = get_trainer_dynamic(bench.build_args(cfg, "cuda:0"))
b = synthetic_batch(2, 160, 160, seed=1)
b = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
for _ in range(3):
    tr.train(b, keep_predictions=False)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.train(b, keep_predictions=False)
    torch.cuda.synchronize()
seen = {}
for ev in prof.events():
    n = ev.name
    if not any(k in n for k in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::cat", "aten::mul", "aten::add", "aten::stack", "aten::ones_like", "aten::zeros", "aten::full", "aten::contiguous", "aten::clone")):
        continue
    st = [s for s in (ev.stack or []) if "site-packages" not in s and "dist-packages" not in s and "<built-in" not in s][:5]
    if not st:
        st = list(ev.stack or [])[:6]
    key = (n, tuple(st))
    seen[key] = seen.get(key, 0) + 1
for (n, st), c in sorted(seen.items(), key=lambda kv: -kv[1]):
    print("%3d x %-18s %s" % (c, n, " <- ".join(s.split("/")[-1][:70] for s in st)))
