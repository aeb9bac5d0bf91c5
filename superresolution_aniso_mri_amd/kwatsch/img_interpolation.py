"""Latent-space interpolation helpers of the reference's kwatsch/img_interpolation.py:20-89 on the HIP engine."""
import numpy as np
import torch

from .. import ops
from .acai_utils import make_grid


def latent_space_interp(alpha, trainer, img1, img2, device=None, hierarchical=False, with_labels=False):
    """decode(alpha*enc(img1) + (1-alpha)*enc(img2)) (reference :57-89)."""
    if hierarchical or with_labels:
        raise NotImplementedError("hierarchical / labelled latents are outside the ae_combined path")
    dev = device or trainer.args["device"]
    z = torch.cat([trainer.encode(img1.float().to(dev), use_sr_model=True), trainer.encode(img2.float().to(dev), use_sr_model=True)])
    inter = trainer.decode(ops.lerp_mix(z, float(alpha), float(1 - alpha)), use_sr_model=True)
    return {"inter_image": inter.detach().cpu().contiguous(), "inter_label": None}


def latent_space_interp_diff_patch_size(alpha, trainer, img1, img2, patch_size, device=None):
    """Patch-tiled variant (reference :20-54): tile each slice into non-overlapping patches, interpolate the patch
    latents with z1*(1-alpha) + z2*alpha, decode and re-assemble.  All slices' patches form ONE batch here."""
    dev = device or trainer.args["device"]
    if img1.dim() == 4:
        img1 = img1.squeeze(dim=1)
    if img2.dim() == 4:
        img2 = img2.squeeze(dim=1)
    S, H, W = img1.shape
    ph, pw = int(patch_size[0]), int(patch_size[1])
    ny, nx = H // ph, W // pw

    def tiles(v):
        v = v.float().to(dev)[:, :ny * ph, :nx * pw].reshape(S, ny, ph, nx, pw).permute(0, 1, 3, 2, 4)
        return v.reshape(S * ny * nx, 1, ph, pw).contiguous()

    z = torch.cat([trainer.encode(tiles(img1), use_sr_model=True), trainer.encode(tiles(img2), use_sr_model=True)])
    rec = trainer.decode(ops.lerp_mix(z, float(1 - alpha), float(alpha)), use_sr_model=True)
    rec = rec.reshape(S, ny, nx, ph, pw).permute(0, 1, 3, 2, 4).reshape(S, 1, ny * ph, nx * pw)
    return rec.detach().cpu().contiguous()


def make_interp_image_grid(interp_images, num_interpolations, normalize=True):
    b = interp_images.size(0)
    num_rows = num_interpolations + 2
    assert b % num_rows == 0
    width = b // num_rows
    imgs = interp_images.float()
    if normalize:
        imgs = (imgs - imgs.min()) / (imgs.max() - imgs.min() + 1e-12)
    return make_grid(imgs, width).numpy(), width, num_rows
