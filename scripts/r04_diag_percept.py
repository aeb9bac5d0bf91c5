#!/usr/bin/env python
"""First-step gradients of the cardiac_percept fixture with the one-launch BatchNorm on / off: per-parameter rel-L2 against the reference vector."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_step as T
tag = sys.argv[1] if len(sys.argv) > 1 else "cardiac_percept"
rec = dict(np.load(os.path.join(T.GOLDEN, "step_k3_%s.npz" % tag)))
res = {}
for fused in ("0", "1"):
    os.environ["AESR_BN_FUSED"] = fused
    tr = T.make_trainer(tag, rec)
    tr.opt_ae.param_groups[0]["lr"] = 0.0
    tr.train(T._batch(rec, 0), keep_predictions=False)
    torch.cuda.synchronize()
    res[fused] = {k: p.grad.detach().cpu().clone() for k, p in tr.model.named_parameters()}
for k in res["0"]:
    ref = torch.from_numpy(rec["grad0/" + k])
    print("%-16s 3-launch %.2e   one-launch %.2e   between %.2e" % (k, T.rel_l2(res["0"][k], ref), T.rel_l2(res["1"][k], ref), T.rel_l2(res["1"][k], res["0"][k])))
