"""Evaluation protocol around the slice synthesis: the reference's ``evaluate/common.py`` on the HIP engine.

``create_super_volume`` (:134-235) is the function the evaluation suites call (``evaluate/create_HR_images.py:309``,
``evaluate/evaluate_image.py:69``): keep every ``downsample_steps``-th slice of a volume, synthesise the slices in between
from the latent mixes, and append the slices that the sub-sampling could not pair.  The synthesis itself is
``generate_hr_volumes.create_super_volume`` (every slice encoded once, all mixes decoded as one batch, interleave and clamp
on the device); this module adds the host-side bookkeeping with the reference's argument names and return keys."""
import os

import numpy as np
import torch

from .. import generate_hr_volumes as _ghv


def save_metrics(output_dir, eval_dataset, metrics_dict, downsample_steps, interpol_method, eval_axis):
    """<output_dir>/results/[<dataset>_]<method>_<k>x[_axis<a>].npz (:11-25); the results directory must not exist twice."""
    output_dir = os.path.join(output_dir, "results")
    if not os.path.isdir(output_dir):
        os.makedirs(output_dir, exist_ok=False)
    name = "{}_{}x.npz".format(interpol_method, downsample_steps) if eval_axis == 0 else \
        "{}_{}x_axis{}.npz".format(interpol_method, downsample_steps, eval_axis)
    if eval_dataset is not None:
        name = "{}_".format(eval_dataset) + name
    path = os.path.join(output_dir, name)
    np.savez(path, **metrics_dict)
    print("INFO - Saved results to {}".format(path))


def strip_conventional_interpolation_results(img3d_sr, img3d_original, expand_factor):
    """Drop the last ``expand_factor`` slices of a conventionally expanded volume and put the last original slice back (:28-32)."""
    return np.concatenate((img3d_sr[:-expand_factor], img3d_original[-1][None]))


def determine_last_slice(orig_num_slices, downsample_steps):
    """Index of the last slice that survives ``[::downsample_steps]`` (:35-38)."""
    return ((int(orig_num_slices) - 1) // int(downsample_steps)) * int(downsample_steps)


def rescale_tensor(p_tensor):
    lo, hi = torch.min(p_tensor), torch.max(p_tensor)
    return torch.clamp((p_tensor - lo) / (hi - lo), min=0, max=1)


def apply_blur_filter(img, sigma=1.):
    """Slice-wise Gaussian blur on the host (:122-126; scipy, used for the blurred conventional baseline only)."""
    from scipy.ndimage import gaussian_filter
    return np.stack([gaussian_filter(img[i], sigma) for i in range(img.shape[0])]) if img.shape[0] else np.zeros_like(img)


def create_recon_from_diff_psize(trainer, test_images, patch_size=(32, 32)):
    """Reconstruct ONE slice [y,x] patch by patch (:52-64): non-overlapping tiles form the batch, the result is the tile
    grid put back together (the part of the slice the tiles cover)."""
    if test_images.dim() > 2:
        test_images = torch.squeeze(test_images)
    return eval_on_different_patch_size(trainer, test_images[None], patch_size)[0]


def eval_on_different_patch_size(trainer, test_images, patch_size=(32, 32)):
    """[z,y,x] CPU tensor -> [z, ny*ph, nx*pw] reconstructions computed on ph x pw tiles (:41-49).  All tiles of all slices
    go through the network as one batch (the reference runs one slice at a time)."""
    if not isinstance(patch_size, tuple):
        patch_size = (int(patch_size), int(patch_size))
    ph, pw = int(patch_size[0]), int(patch_size[1])
    Z, H, W = test_images.shape
    ny, nx = H // ph, W // pw
    dev = trainer.args["device"]
    tiles = test_images.float().to(dev)[:, :ny * ph, :nx * pw].reshape(Z, ny, ph, nx, pw).permute(0, 1, 3, 2, 4)
    rec = trainer.predict(tiles.reshape(Z * ny * nx, 1, ph, pw).contiguous())
    rec = rec.reshape(Z, ny, nx, ph, pw).permute(0, 1, 3, 2, 4).reshape(Z, ny * ph, nx * pw)
    return rec.detach().cpu().contiguous()


def create_super_volume(trainer, images, alpha_range=None, use_original=False, hierarchical=False, downsample_steps=None,
                        generate_inbetween_slices=False, train_patch_size=None, feature_dict=None, labels=None):
    """images [z,y,x] -> {'upsampled_image' [z',y,x] (CPU, clamped to [0,1]), 'upsampled_labels' None, 'pred_alphas'
    [(z_kept-1)*n, 1, y, x]} with the reference's conventions (:134-235):

    * ``generate_inbetween_slices`` without ``downsample_steps`` sub-samples by ``len(alpha_range)+1``;
    * the last ``(z-1) % downsample_steps`` slices cannot be paired: they are cut before sub-sampling and, when in-between
      slices are generated, the ORIGINAL slices are appended after the synthesised stack;
    * ``alpha_range`` defaults to (0.25, 0.5, 0.75); alpha weights the LATER slice of each pair."""
    if hierarchical or labels is not None:
        raise NotImplementedError("hierarchical latents / label channels are outside the ae_combined path")
    if images.dim() != 3:
        raise ValueError("create_super_volume expects a [z, y, x] volume, got shape %s" % (tuple(images.shape),))
    if generate_inbetween_slices and downsample_steps is None:
        downsample_steps = int(len(alpha_range) + 1)
    orig, remain = images, 0
    if not generate_inbetween_slices and downsample_steps is not None:
        print("WARNING !!! create_super_volume - downsample steps is not None but generate_inbetween_slices is False!")
    if downsample_steps is not None or generate_inbetween_slices:
        remain = (orig.shape[0] - 1) % downsample_steps
        if remain:
            images = images[:-remain]
        images = images[::downsample_steps]
    if alpha_range is None:
        alpha_range = [0.25, 0.5, 0.75]
    out = _ghv.create_super_volume(trainer, images, alpha_range, use_original=use_original)
    new_volume = out["upsampled_image"]
    if generate_inbetween_slices and remain:
        new_volume = torch.clamp(torch.cat([new_volume, orig[-remain:].float().cpu()]), min=0, max=1.)
    pairs = images.shape[0] - 1
    alphas = torch.tensor([float(a) for a in alpha_range], dtype=torch.float32).repeat_interleave(pairs)
    pred_alphas = alphas[:, None, None, None].expand(-1, 1, images.shape[1], images.shape[2])
    return {"upsampled_image": new_volume, "upsampled_labels": None, "pred_alphas": pred_alphas}


def create_simple_interpolation(images, spacing, new_spacing_z=None, expand_factor=None, interpol_filter=None,
                                generate_inbetween_slices=False):
    """The conventional (Lanczos / B-spline / linear) through-plane baseline is SimpleITK's ExpandImageFilter in the reference
    (:76-118): a comparison method, not part of the synthesis path."""
    raise NotImplementedError("conventional interpolation baselines (SimpleITK ExpandImageFilter) are outside this build")
