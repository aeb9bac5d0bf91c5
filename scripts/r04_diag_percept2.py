#!/usr/bin/env python
"""How far does the first-step gradient of the cardiac_percept fixture move under roundings that are all equally valid?  Variants:
three-launch BatchNorm + Winograd kernels (round 3's path), one-launch BatchNorm, direct fp32 kernels (AESR_WINO=0 AESR_WGRAD_WINO=0),
and the three-launch path with the INPUT perturbed by one part in 1e7.  Run each in a child process (the switches are read at import)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import test_gpu_step as T
tag = "cardiac_percept"
rec = dict(np.load(os.path.join(T.GOLDEN, "step_k3_%%s.npz" %% tag)))
tr = T.make_trainer(tag, rec)
tr.opt_ae.param_groups[0]["lr"] = 0.0
b = T._batch(rec, 0)
eps = float(os.environ.get("PERTURB", "0"))
if eps:
    b["image"] = b["image"] * (1.0 + eps)
tr.train(b, keep_predictions=False)
torch.cuda.synchronize()
worst = max(T.rel_l2(p.grad, rec["grad0/" + k]) for k, p in tr.model.named_parameters())
med = float(np.median([T.rel_l2(p.grad, rec["grad0/" + k]) for k, p in tr.model.named_parameters()]))
print("%%-58s worst %%.2e  median %%.2e  loss %%.8f" %% (os.environ["VARIANT"], worst, med, float(tr.losses["loss_ae"][-1])))
''' % (ROOT, ROOT)
for name, env in [("three-launch BN, Winograd (round 3 path)", {"AESR_BN_FUSED": "0"}),
                  ("one-launch BN, Winograd", {"AESR_BN_FUSED": "1"}),
                  ("three-launch BN, direct fp32 kernels", {"AESR_BN_FUSED": "0", "AESR_WINO": "0", "AESR_WGRAD_WINO": "0"}),
                  ("one-launch BN, direct fp32 kernels", {"AESR_BN_FUSED": "1", "AESR_WINO": "0", "AESR_WGRAD_WINO": "0"}),
                  ("three-launch BN, Winograd, input x (1 + 1e-7)", {"AESR_BN_FUSED": "0", "PERTURB": "1e-7"}),
                  ("three-launch BN, Winograd, input x (1 - 1e-7)", {"AESR_BN_FUSED": "0", "PERTURB": "-1e-7"}),
                  ("three-launch BN, Winograd, stem unfolded", {"AESR_BN_FUSED": "0", "AESR_FUSE_STEM": "0"}),
                  ("one-launch BN, Winograd, input x (1 + 1e-7)", {"AESR_BN_FUSED": "1", "PERTURB": "1e-7"}),
                  ("one-launch BN, Winograd, input x (1 - 1e-7)", {"AESR_BN_FUSED": "1", "PERTURB": "-1e-7"}),
                  ("one-launch BN, Winograd, input x (1 + 3e-7)", {"AESR_BN_FUSED": "1", "PERTURB": "3e-7"}),
                  ("three-launch BN, Winograd, input x (1 + 3e-7)", {"AESR_BN_FUSED": "0", "PERTURB": "3e-7"}),
                  ("three-launch BN, Winograd, input x (1 - 3e-7)", {"AESR_BN_FUSED": "0", "PERTURB": "-3e-7"}),
                  ("one-launch BN, Winograd, stem unfolded", {"AESR_BN_FUSED": "1", "AESR_FUSE_STEM": "0"})]:
    e = dict(os.environ, VARIANT=name, **env)
    subprocess.run([sys.executable, "-c", CHILD], env=e, stderr=subprocess.DEVNULL)
