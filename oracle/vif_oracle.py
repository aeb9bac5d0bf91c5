"""Oracle (TEST INFRASTRUCTURE, see oracle/__init__.py): the VIF score of the reference's evaluation protocol, restated in numpy.

Follows
  * evaluate/vifvec.py:7-63      ``vifp_mscale`` (pixel-domain VIF over four scales)
  * evaluate/metrics.py:65-109   ``compute_vif_for_batch``: BOTH volumes are converted to uint8 (``np.uint8(np.clip(x * 255., 0, 255))``,
                                 :72-73) BEFORE ``vifp_mscale`` is called per slice, the mean is taken over the slices that are kept
                                 (synthesised slices when ``downsample_steps`` is given) and whose score is finite

What that means for the arithmetic (and what the device kernel csrc/vif.hip has to reproduce):
  * ``scipy.ndimage.gaussian_filter`` returns its INPUT dtype.  On uint8 images every 1-D pass computes in double and stores the
    result back as uint8 by a C cast, i.e. TRUNCATED toward zero, after the first axis and again after the second;
  * ``ref * ref``, ``mu1 * mu1``, ``gaussian_filter(ref * ref) - mu1_sq`` ... are uint8 expressions: they wrap modulo 256;
    ``sigma1_sq[sigma1_sq < 0] = 0`` never fires;
  * from ``g = sigma12 / (sigma1_sq + eps)`` on everything is float64.
The reference evaluates exactly this (its model-selection files carry these numbers), so this oracle does too -- dtype by dtype.  The
same function on float input (``vifp_mscale`` called directly) follows the same lines with float arithmetic.

The Gaussian filter is restated here instead of calling scipy so that the summation order the kernel must follow is explicit:
scipy's NI_Correlate1D, symmetric-weights branch:  tmp = x[l] w[c];  for ii = -r .. -1:  tmp += (x[l + ii] + x[l - ii]) w[ii + r]
with the 'reflect' boundary (d c b a | a b c d | d c b a), weights = ``_gaussian_kernel1d(sigma, 0, int(4 sigma + 0.5))``.
tests/test_oracle_golden.py pins this file against tests/golden/vif.npz = outputs of the reference's own functions (which call scipy)."""
import numpy as np

EPS = 1e-10


def gaussian_weights(sd, truncate=4.0):
    """(weights[2 r + 1] float64, r): scipy.ndimage._filters._gaussian_kernel1d(sd, 0, r) with r = int(truncate * sd + 0.5)."""
    r = int(truncate * float(sd) + 0.5)
    x = np.arange(-r, r + 1)
    phi = np.exp(-0.5 / (sd * sd) * x ** 2)
    return phi / phi.sum(), r


def scale_sigmas():
    """sd of the four scales (evaluate/vifvec.py:17-18): N = 17, 9, 5, 3; sd = N / 5."""
    return [(2 ** (4 - s + 1) + 1) / 5.0 for s in range(1, 5)]


def correlate1d_reflect(a, w, r, axis):
    """One pass of the filter along ``axis``; the result has the dtype of ``a`` (integers: truncated, as the C cast does)."""
    x = np.moveaxis(np.asarray(a), axis, -1)
    n = x.shape[-1]
    pad = np.pad(x.astype(np.float64), [(0, 0)] * (x.ndim - 1) + [(r, r)], mode="symmetric")      # numpy 'symmetric' == scipy 'reflect'
    out = pad[..., r:r + n] * w[r]
    for ii in range(-r, 0):
        out = out + (pad[..., r + ii:r + ii + n] + pad[..., r - ii:r - ii + n]) * w[ii + r]
    return np.moveaxis(out.astype(a.dtype), -1, axis)


def gaussian_filter(a, sd):
    w, r = gaussian_weights(sd)
    return correlate1d_reflect(correlate1d_reflect(a, w, r, 0), w, r, 1)


def vifp_mscale(ref, dist, sigma_nsq=2.0):
    """evaluate/vifvec.py:7-63, line by line (``do_rescale`` left out: no caller sets it); dtype semantics are numpy's, i.e. the reference's."""
    num, den = 0.0, 0.0
    for scale, sd in zip(range(1, 5), scale_sigmas()):
        if scale > 1:
            ref = gaussian_filter(ref, sd)[::2, ::2]
            dist = gaussian_filter(dist, sd)[::2, ::2]
        mu1, mu2 = gaussian_filter(ref, sd), gaussian_filter(dist, sd)
        mu1_sq, mu2_sq, mu1_mu2 = mu1 * mu1, mu2 * mu2, mu1 * mu2
        sigma1_sq = gaussian_filter(ref * ref, sd) - mu1_sq
        sigma2_sq = gaussian_filter(dist * dist, sd) - mu2_sq
        sigma12 = gaussian_filter(ref * dist, sd) - mu1_mu2
        sigma1_sq[sigma1_sq < 0] = 0
        sigma2_sq[sigma2_sq < 0] = 0
        g = sigma12 / (sigma1_sq + EPS)
        sv_sq = sigma2_sq - g * sigma12
        g[sigma1_sq < EPS] = 0
        sv_sq[sigma1_sq < EPS] = sigma2_sq[sigma1_sq < EPS]
        sigma1_sq[sigma1_sq < EPS] = 0
        g[sigma2_sq < EPS] = 0
        sv_sq[sigma2_sq < EPS] = 0
        sv_sq[g < 0] = sigma2_sq[g < 0]
        g[g < 0] = 0
        sv_sq[sv_sq <= EPS] = EPS
        sigma1_sq = sigma1_sq.astype(np.float64)
        num += np.sum(np.log10(1 + g * g * sigma1_sq / (sv_sq + sigma_nsq)))
        den += np.sum(np.log10(1 + sigma1_sq / sigma_nsq))
    return num / den if den != 0 else np.nan


def to_uint8(x):
    """evaluate/metrics.py:72-73 (the images arrive as float32, :48-55)."""
    return np.uint8(np.clip(np.asarray(x, dtype=np.float32) * 255., 0, 255))


def original_slice_ids(n, downsample_steps, conv_interpol=False):
    """evaluate/metrics.py:29-45: the slices of an up-sampled volume that are originals (skipped when scoring)."""
    ids = np.arange(n)
    keep = None
    if (n - 1) % downsample_steps != 0:
        rem = (n - 1) % downsample_steps
        keep, ids = ids[-rem:], ids[:-rem]
    if conv_interpol and ids.shape[0] % downsample_steps != 0:
        rem = ids.shape[0] % downsample_steps
        keep = ids[-rem:] if keep is None else np.concatenate((ids[-rem:], keep))
        ids = ids[:-rem]
    ids = ids[::downsample_steps]
    return ids if keep is None else np.concatenate((ids, keep))


def compute_vif_for_batch(images, recons, downsample_steps=None, conv_interpol=False):
    """evaluate/metrics.py:65-109 for eval_axis = 0, normalize = False: (mean over the kept, finite slices; per-slice scores)."""
    a, b = np.squeeze(to_uint8(images)), np.squeeze(to_uint8(recons))
    if a.ndim == 2:
        v = vifp_mscale(a, b)
        return v, np.array([v])
    skip = set(original_slice_ids(a.shape[0], downsample_steps, conv_interpol).tolist()) if downsample_steps is not None else set()
    per = np.array([np.nan if z in skip else vifp_mscale(a[z], b[z]) for z in range(a.shape[0])])
    ok = np.isfinite(per)
    return (float(np.mean(per[ok])) if ok.any() else float("nan")), per
