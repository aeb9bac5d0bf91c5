"""Validation / model-selection metrics on the device: the reference's ``evaluate/metrics.py`` SSIM and PSNR entry points
(:111-154 ``compute_ssim_for_batch``, :157-194 ``compute_psnr_for_batch``, :29-45 ``determine_original_sliceids``) with the
per-slice skimage calls replaced by ONE HIP reduction over the whole volume (``aesr_ssim_mse``).

Definitions (skimage ``structural_similarity`` / ``peak_signal_noise_ratio`` defaults): uniform 7x7 window (5x5 when a slice
is smaller than 8 pixels in a dimension, as the reference does for long-axis views), K1 = 0.01, K2 = 0.03, sample
covariance, mean over the fully covered windows; PSNR = 10 log10(R^2 / mse).  ``data_range`` R is explicit here
(default 1.0: images are normalised to [0, 1]); skimage releases older than 0.19 silently used R = 2 for float input --
pass ``data_range=2.0`` to reproduce numbers produced with such an installation.
``compute_lpips_for_batch`` (:210-243) scores all kept slices with ONE batched LPIPS pass instead of a criterion call and a host
sync per slice.  ``compute_vif_for_batch`` (:65-109 -> evaluate/vifvec.py:7-63 ``vifp_mscale``) scores all slices of a volume with
one device pass (``aesr_vif_mscale``, csrc/vif.hip) in the reference's own arithmetic: uint8 images, uint8 Gaussian filter, products
modulo 256 -- see oracle/vif_oracle.py.  Not covered: the HD metrics of the same file (outside the ae_combined path)."""
import numpy as np
import torch

from .. import _hip
from .._hip import check, lib, ptr, stream


def determine_original_sliceids(reference, downsample_steps, conv_interpol=False):
    """Indices of the slices that are ORIGINAL (not synthesised) in a volume that was sub-sampled by ``downsample_steps`` and
    up-sampled again (evaluate/metrics.py:29-45): they are skipped when scoring."""
    n = reference.shape[0]
    slice_ids = np.arange(n)
    keep = None
    if (n - 1) % downsample_steps != 0:
        rem = (n - 1) % downsample_steps
        keep = slice_ids[-rem:]
        slice_ids = slice_ids[:-rem]
    if conv_interpol and slice_ids.shape[0] % downsample_steps != 0:
        rem = slice_ids.shape[0] % downsample_steps
        keep = slice_ids[-rem:] if keep is None else np.concatenate((slice_ids[-rem:], keep))
        slice_ids = slice_ids[:-rem]
    slice_ids = slice_ids[::downsample_steps]
    if keep is not None:
        slice_ids = np.concatenate((slice_ids, keep))
    return slice_ids


def _as_volume(t, device):
    """numpy / torch, [b,c,h,w] or [z,h,w] or [h,w] -> contiguous fp32 CUDA [Z,H,W] (evaluate/metrics.py:48-65 squeeze rule)."""
    if isinstance(t, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(t))
    t = t.detach().to(device=device, dtype=torch.float32)
    if t.dim() > 3:
        t = t.squeeze()
    if t.dim() == 2:
        t = t.unsqueeze(0)
    if t.dim() != 3:
        raise ValueError("expected an image, a volume or a [b,1,h,w] batch, got shape %s" % (tuple(t.shape),))
    return t.contiguous()


def rescale_intensities(vol, percs=(0, 100)):
    """datasets/common.py:408-417 on the device."""
    q = torch.quantile(vol.flatten().double(), torch.tensor([percs[0] / 100.0, percs[1] / 100.0], dtype=torch.float64, device=vol.device))
    lo, hi = float(q[0]), float(q[1])
    return ((vol - lo) / (hi - lo)).clamp(0, 1)


def slice_ssim_psnr(l_images, l_reconstructions, data_range=1.0, win_size=None, device="cuda"):
    """Per-slice (ssim[Z], psnr[Z], mse[Z]) as float64 numpy arrays; one device pass over both volumes."""
    a, b = _as_volume(l_images, device), _as_volume(l_reconstructions, device)
    if a.shape != b.shape:
        raise ValueError("shape mismatch %s vs %s" % (tuple(a.shape), tuple(b.shape)))
    Z, H, W = a.shape
    if win_size is None:
        win_size = 7 if min(H, W) >= 8 else 5          # evaluate/metrics.py:141-147
    _hip.require_gpu_tensor(a, "images")
    ws = torch.empty(lib.aesr_ssim_workspace_doubles(Z, H, W), device=a.device, dtype=torch.float64)
    out = torch.empty((2, Z), device=a.device, dtype=torch.float64)
    check(lib.aesr_ssim_mse(ptr(a), ptr(b), ptr(ws), ptr(out[0]), ptr(out[1]), Z, H, W, int(win_size), float(data_range), 0.01,
                            0.03, stream()), "aesr_ssim_mse")
    ssim, mse = out[0].cpu().numpy(), out[1].cpu().numpy()
    with np.errstate(divide="ignore"):
        psnr = 10.0 * np.log10(float(data_range) ** 2 / mse)
    return ssim, psnr, mse


def gaussian_kernel1d(sd, truncate=4.0):
    """(weights float64 [2 r + 1], r) of scipy.ndimage.gaussian_filter(., sd): ``_gaussian_kernel1d(sd, 0, int(truncate * sd + 0.5))``,
    computed with numpy exactly as scipy does -- the uint8 filter TRUNCATES its result, so wherever an image is constant the last bit
    of a weight decides between v and v - 1; the kernel therefore takes the weights from here instead of recomputing them."""
    r = int(truncate * float(sd) + 0.5)
    x = np.arange(-r, r + 1)
    phi = np.exp(-0.5 / (sd * sd) * x ** 2)
    return phi / phi.sum(), r


_VIF_FILTERS = None


def _vif_filters():
    global _VIF_FILTERS
    if _VIF_FILTERS is None:
        ks = [gaussian_kernel1d((2 ** (4 - s + 1) + 1) / 5.0) for s in range(1, 5)]       # evaluate/vifvec.py:17-18: N = 17, 9, 5, 3; sd = N / 5
        _VIF_FILTERS = (_hip.double_array(np.concatenate([k[0] for k in ks])), _hip.int_array([k[1] for k in ks]))
    return _VIF_FILTERS


def slice_vif(l_images, l_reconstructions, sigma_nsq=2.0, device="cuda"):
    """Per-slice VIF [Z] (float64 numpy; NaN where the reference slice is black): ``vifp_mscale`` of the uint8 images, one device pass."""
    a, b = _as_volume(l_images, device), _as_volume(l_reconstructions, device)
    if a.shape != b.shape:
        raise ValueError("shape mismatch %s vs %s" % (tuple(a.shape), tuple(b.shape)))
    _hip.require_gpu_tensor(a, "images")
    Z, H, W = a.shape
    w, r = _vif_filters()
    out = np.empty(Z, dtype=np.float64)
    for z0 in range(0, Z, 4096):
        n = min(4096, Z - z0)
        ws = torch.empty(lib.aesr_vif_workspace_bytes(n, H, W), device=a.device, dtype=torch.uint8)
        vif = torch.empty(n, device=a.device, dtype=torch.float64)
        check(lib.aesr_vif_mscale(ptr(a[z0:z0 + n]), ptr(b[z0:z0 + n]), ptr(ws), ptr(vif), n, H, W, w, r, float(sigma_nsq), stream()),
              "aesr_vif_mscale")
        out[z0:z0 + n] = vif.cpu().numpy()
    return out


def compute_vif_for_batch(l_images, l_reconstructions, eval_axis=0, normalize=False, downsample_steps=None, conv_interpol=False,
                          device="cuda"):
    """Mean VIF over the slices of a volume whose score is finite (original slices skipped when ``downsample_steps`` is given); a single
    2-D image returns its score.  evaluate/metrics.py:65-109."""
    _check_axis(eval_axis)
    single = (torch.as_tensor(l_images).squeeze().dim() == 2) if not isinstance(l_images, np.ndarray) else (np.squeeze(l_images).ndim == 2)
    a, b = _as_volume(l_images, device), _as_volume(l_reconstructions, device)
    if normalize:
        b = rescale_intensities(b, percs=(0, 100))
    vif = slice_vif(a, b, device=device)
    if single:
        return float(vif[0])
    keep = np.isfinite(vif)
    if downsample_steps is not None:
        keep[determine_original_sliceids(a, downsample_steps, conv_interpol)] = False
    return float(np.mean(vif[keep])) if keep.any() else float("nan")


def _check_axis(eval_axis):
    if eval_axis != 0:
        raise NotImplementedError("long-axis (eval_axis != 0) evaluation is outside the ae_combined path")


def compute_ssim_for_batch(l_images, l_reconstructions, eval_axis=0, normalize=False, downsample_steps=None, conv_interpol=False,
                           data_range=1.0, device="cuda"):
    """Mean SSIM over the slices of a volume (original slices skipped when ``downsample_steps`` is given), evaluate/metrics.py:111-154."""
    _check_axis(eval_axis)
    a, b = _as_volume(l_images, device), _as_volume(l_reconstructions, device)
    if normalize:
        b = rescale_intensities(b, percs=(0, 100))
    ssim, _, _ = slice_ssim_psnr(a, b, data_range=data_range, device=device)
    keep = np.ones(a.shape[0], dtype=bool)
    if downsample_steps is not None and a.shape[0] > 1:
        keep[determine_original_sliceids(a, downsample_steps, conv_interpol)] = False
    return float(np.mean(ssim[keep]))


def compute_psnr_for_batch(l_images, l_reconstructions, eval_axis=0, normalize=False, downsample_steps=None, conv_interpol=False,
                           data_range=1.0, device="cuda"):
    """Mean PSNR over the slices (nan / inf slices dropped as in the reference), evaluate/metrics.py:157-194."""
    _check_axis(eval_axis)
    a, b = _as_volume(l_images, device), _as_volume(l_reconstructions, device)
    if normalize:
        b = rescale_intensities(b, percs=(0, 100))
    _, psnr, _ = slice_ssim_psnr(a, b, data_range=data_range, device=device)
    keep = np.isfinite(psnr)
    if downsample_steps is not None and a.shape[0] > 1:
        keep[determine_original_sliceids(a, downsample_steps, conv_interpol)] = False
    if a.shape[0] == 1:
        return float(psnr[0])
    return float(np.mean(psnr[keep]))


def compute_lpips_for_batch(l_images, l_reconstructions, eval_axis=0, normalize=False, downsample_steps=None, conv_interpol=False,
                            criterion=None, device="cuda"):
    """Mean LPIPS distance over the slices of a volume (original slices skipped when ``downsample_steps`` is given), reference
    evaluate/metrics.py:210-243.  ``criterion``: a ``PerceptualLoss`` (built with the reference's arguments when omitted)."""
    _check_axis(eval_axis)
    if criterion is None:
        from ..lpips.perceptual import PerceptualLoss
        criterion = PerceptualLoss(model="net-lin", net="vgg", use_gpu=True, gpu_ids=[0], device=device)
    a, b = _as_volume(l_images, device), _as_volume(l_reconstructions, device)
    if normalize:
        b = rescale_intensities(b, percs=(0, 100))
    keep = np.ones(a.shape[0], dtype=bool)
    if downsample_steps is not None and a.shape[0] > 1:
        keep[determine_original_sliceids(a, downsample_steps, conv_interpol)] = False
    idx = torch.from_numpy(np.nonzero(keep)[0]).to(a.device)
    with torch.no_grad():
        d = criterion(a[idx][:, None].contiguous(), b[idx][:, None].contiguous(), normalize=True)
    return float(d.double().mean())
