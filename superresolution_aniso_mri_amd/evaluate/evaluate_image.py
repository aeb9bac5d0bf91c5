"""Per-image evaluation helpers of the reference's ``evaluate/evaluate_image.py`` on the HIP path: ``evaluate_image`` (:37-80,
held-out-slice synthesis of every frame of a 4-D image with alpha = 0.5), ``compute_stats`` (:11-34, SSIM / PSNR / LPIPS of a
volume) and ``create_compare_image`` (:83-106, the slice-by-slice comparison grid)."""
import numpy as np
import torch

from ..kwatsch.acai_utils import make_grid
from .common import create_super_volume
from .find_best_model import get_transforms
from .metrics import compute_psnr_for_batch, compute_ssim_for_batch, compute_vif_for_batch


def compute_stats(trainer, original_img, r_s_img, normalize=True, downsample_steps=None, is_conv_meth=False):
    """(ssim, psnr, vif, lpips) of a reconstructed / synthesised volume against the original [z,y,x].  As in the reference the
    LPIPS term is taken over every ``downsample_steps``-th slice pair (``[::downsample_steps]``)."""
    ssim_res = compute_ssim_for_batch(original_img, r_s_img, eval_axis=0, downsample_steps=downsample_steps, conv_interpol=is_conv_meth,
                                      normalize=normalize)
    psnr_res = compute_psnr_for_batch(original_img, r_s_img, eval_axis=0, downsample_steps=downsample_steps, conv_interpol=is_conv_meth,
                                      normalize=normalize)
    vif_res = compute_vif_for_batch(original_img, r_s_img, eval_axis=0, downsample_steps=downsample_steps, conv_interpol=is_conv_meth,
                                    normalize=normalize)                       # reference :19-21
    dev = trainer.args["device"]
    a = torch.as_tensor(original_img, dtype=torch.float32).to(dev)[:, None]
    b = torch.as_tensor(r_s_img, dtype=torch.float32).to(dev)[:, None]
    if downsample_steps is not None:
        a, b = a[::downsample_steps], b[::downsample_steps]
    if trainer.percept_criterion is None:
        raise ValueError("compute_stats needs a trainer with a perceptual criterion (trainer.percept_criterion is None)")
    with torch.no_grad():
        lpips_res = float(trainer.percept_criterion(a.contiguous(), b.contiguous(), normalize=True).mean())
    return ssim_res, psnr_res, vif_res, lpips_res


def evaluate_image(trainer, data_dict, frame_id=None, eval_patch_size=128, downsample_steps=2, transform=None):
    """data_dict: {'image': [t,z,y,x] numpy, 'patient_id', 'spacing', ...}.  Every frame (or ``frame_id``) is padded / centre-cropped
    to ``eval_patch_size``, sub-sampled by ``downsample_steps`` and re-synthesised with alpha = 0.5.
    Returns {'orig_images': {f: [z,y,x]}, 'synth_images': {f: [z,y,x]}, 'pred_alphas': {f: tensor}}."""
    if transform is None:
        transform = get_transforms(eval_patch_size, to_tensor=False)
    images4d = data_dict["image"]
    num_frames = images4d.shape[0]
    if frame_id is None:
        f_range = np.arange(0, num_frames)
    else:
        frame_id = min(frame_id, num_frames - 1)
        f_range = np.arange(frame_id, frame_id + 1)
    synth, orig, alphas = dict(), dict(), dict()
    for f_id in f_range:
        vol = torch.from_numpy(np.ascontiguousarray(transform({"image": images4d[f_id]})["image"]))
        out = create_super_volume(trainer, vol, alpha_range=[0.5], use_original=False, hierarchical=False,
                                  downsample_steps=downsample_steps, generate_inbetween_slices=True)
        synth[f_id] = out["upsampled_image"].detach().cpu().squeeze().numpy()
        orig[f_id] = vol.detach().cpu().squeeze().numpy()
        alphas[f_id] = out["pred_alphas"].squeeze()
    return {"orig_images": orig, "synth_images": synth, "pred_alphas": alphas}


def create_compare_image(real_img, synth_img, downsample_steps=2):
    """Grid (numpy, [1, rows, cols]) of: left neighbours, their reconstructions, held-out slices, synthesised slices, difference,
    reconstructed right neighbours, right neighbours -- one column per slice pair.  Both inputs [#slices, y, x]."""
    if real_img.shape[0] % downsample_steps == 0:
        num_slices = real_img.shape[0] - 1
        real_img, synth_img = real_img[:-1], synth_img[:-1]
    else:
        num_slices = real_img.shape[0]
    s_mask = np.ones(num_slices, dtype=bool)
    s_mask[::downsample_steps] = False
    r_mask = ~s_mask
    slices1, slices3 = real_img[r_mask][:-1], real_img[r_mask][1:]
    rec1, rec3 = synth_img[r_mask][:-1], synth_img[r_mask][1:]
    synth, real = synth_img[s_mask], real_img[s_mask]
    grid = np.concatenate([slices1[:, None], rec1[:, None], real[:, None], synth[:, None], (real - synth)[:, None], rec3[:, None],
                           slices3[:, None]], axis=0)
    return make_grid(torch.from_numpy(np.ascontiguousarray(grid)), slices1.shape[0], padding=2, pad_value=0.5).numpy()
