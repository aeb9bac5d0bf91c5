#!/usr/bin/env python
"""cardiac_percept fixture, one process: the forward tensors and BatchNorm buffers of the one-launch BatchNorm against the three-launch one."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_step as T
tag = sys.argv[1] if len(sys.argv) > 1 else "cardiac_percept"
rec = dict(np.load(os.path.join(T.GOLDEN, "step_k3_%s.npz" % tag)))
out = {}
for fused in ("0", "1"):
    os.environ["AESR_BN_FUSED"] = fused
    tr = T.make_trainer(tag, rec)
    tr.opt_ae.param_groups[0]["lr"] = 0.0
    tr.train(T._batch(rec, 0), keep_predictions=True)
    torch.cuda.synchronize()
    d = {k: v.clone() for k, v in tr.train_predictions.items() if torch.is_tensor(v)}
    d.update({"buf/" + k: v.detach().cpu().clone() for k, v in tr.model.state_dict().items() if "running" in k})
    d.update({"grad/" + k: p.grad.detach().cpu().clone() for k, p in tr.model.named_parameters()})
    out[fused] = d
for k in out["0"]:
    a, b = out["0"][k].double(), out["1"][k].double()
    print("%-28s max|diff| %.3e  rel %.3e  differing elements %d / %d" % (k, float((a - b).abs().max()), float((a - b).norm() / (a.norm() + 1e-30)), int((a != b).sum()), a.numel()))
