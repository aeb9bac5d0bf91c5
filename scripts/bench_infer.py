#!/usr/bin/env python
"""Inference throughput of the slice-synthesis loop (generate_hr_volumes.create_super_volume): synthesised slices per second on a
synthetic volume, random-init weights.  bench_infer.py [Z H num_interpolations]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superresolution_aniso_mri_amd import generate_hr_volumes as ghv  # noqa: E402
from superresolution_aniso_mri_amd.kwatsch.get_trainer import get_trainer_dynamic  # noqa: E402
from superresolution_aniso_mri_amd.networks.net_config import NetworkConfig  # noqa: E402

Z, H, n = (int(v) for v in (sys.argv[1:4] + ["30", "160", "6"][len(sys.argv) - 1:]))
args = dict(model="ae_combined", dataset="ACDC", device="cuda", lr=1e-5, weight_decay=0.0, epochs=1, width=128, latent_width=32, depth=32,
            latent=128, ex_loss_weight1=0.05, use_percept_loss=False, get_masks=False, use_loss_annealing=False,
            use_extra_latent_loss=False, epoch_threshold=0, ae_class="VanillaACAI", image_mix_loss_func="mse")
for k, v in NetworkConfig("ae_combined", dataset="ACDC").architecture.items():
    args.setdefault(k, v)
torch.manual_seed(0)
tr = get_trainer_dynamic(args, eval_mode=True)
vol = torch.rand(Z, H, H)
alphas = np.linspace(0, 1, n + 2, endpoint=True)[1:-1]
ghv.create_super_volume(tr, vol, alphas, use_original=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    out = ghv.create_super_volume(tr, vol, alphas, use_original=True)["upsampled_image"]
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
synth = (Z - 1) * n
print("volume %dx%dx%d, %d interpolations: %d synthesised slices in %.1f ms = %.0f slices/s (incl. the device-to-host copy of %d slices)"
      % (Z, H, H, n, synth, dt * 1e3, synth / dt, out.shape[0]))
