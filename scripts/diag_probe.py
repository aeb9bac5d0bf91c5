#!/usr/bin/env python
"""ae_acdc_probe (2 images of 160x160, the full ACDC model): worst / median relative deviation of the per-parameter gradient norms from the
reference's, one-launch against three-launch BatchNorm (child processes: kernel switches are read at import)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import test_gpu_ae as T
from oracle import step_oracle
rec = dict(np.load(os.path.join(T.GOLDEN, "ae_acdc_probe.npz")))
torch.manual_seed(892372)
model = T._model("VanillaACAI", dict(width=128, latent_width=32, depth=32, latent=128, colors=1, use_batchnorm=True, use_sigmoid=True, device="cpu")).cuda()
model.train()
image, _ = step_oracle.synthetic_triplets(1, 160, 160, seed=892372)
image = image.cuda()
out = model.decode(model.encode(image))
loss = F.mse_loss(out, image); loss.backward()
dev = {k: abs(p.grad.double().norm().item() - float(rec["gnorm/" + k])) / float(rec["gnorm/" + k]) for k, p in model.named_parameters()}
w = max(dev, key=dev.get)
print("%%-44s worst %%.2e (%%s)  median %%.2e  loss rel %%.1e" %% (os.environ["VARIANT"], dev[w], w, float(np.median(list(dev.values()))), abs(loss.item() - float(rec["loss"])) / float(rec["loss"])))
''' % (ROOT, ROOT)
for name, env in [("three-launch BN", {"AESR_BN_FUSED": "0"}), ("one-launch BN", {"AESR_BN_FUSED": "1"}),
                  ("three-launch BN, AESR_WINO_RING=0", {"AESR_BN_FUSED": "0", "AESR_WINO_RING": "0"}), ("one-launch BN, AESR_WINO_RING=0", {"AESR_BN_FUSED": "1", "AESR_WINO_RING": "0"}),
                  ("three-launch BN, AESR_WINO_RING=2", {"AESR_BN_FUSED": "0", "AESR_WINO_RING": "2"}), ("one-launch BN, AESR_WINO_RING=2", {"AESR_BN_FUSED": "1", "AESR_WINO_RING": "2"}),
                  ("three-launch BN, direct kernels", {"AESR_BN_FUSED": "0", "AESR_WINO": "0", "AESR_WGRAD_WINO": "0"}), ("one-launch BN, direct kernels", {"AESR_BN_FUSED": "1", "AESR_WINO": "0", "AESR_WGRAD_WINO": "0"}),
                  ("three-launch BN, AESR_WINO_RES=0", {"AESR_BN_FUSED": "0", "AESR_WINO_RES": "0"}), ("one-launch BN, AESR_WINO_RES=0", {"AESR_BN_FUSED": "1", "AESR_WINO_RES": "0"})]:
    subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, VARIANT=name, **env), stderr=subprocess.DEVNULL)
