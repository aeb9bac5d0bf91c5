"""Oracle (TEST INFRASTRUCTURE, see oracle/__init__.py): the Laplacian-pyramid L1 loss of kwatsch/lap_pyramid_loss.py restated with
plain torch-CPU ops.  conv_gauss :37-40 (reflect pad 2, depthwise 5x5 binomial /256), downsample :23-24, upsample :27-34
(zeros at the odd rows/columns, then 4*G), laplacian_pyramid :43-53, LapLoss.forward :61-65 (sum of per-level L1 means)."""
import torch
import torch.nn.functional as F


def gauss_kernel(channels, dtype=torch.float32):
    k1 = torch.tensor([1., 4., 6., 4., 1.], dtype=dtype)
    return (torch.outer(k1, k1) / 256.).repeat(channels, 1, 1, 1)


def conv_gauss(img, kernel):
    return F.conv2d(F.pad(img, (2, 2, 2, 2), mode="reflect"), kernel, groups=img.shape[1])


def upsample(x):
    up = torch.zeros(x.shape[0], x.shape[1], 2 * x.shape[2], 2 * x.shape[3], dtype=x.dtype)
    up[:, :, ::2, ::2] = x
    return conv_gauss(up, 4 * gauss_kernel(x.shape[1], x.dtype))


def laplacian_pyramid(img, max_levels=3):
    kernel = gauss_kernel(img.shape[1], img.dtype)
    cur, pyr = img, []
    for _ in range(max_levels):
        down = conv_gauss(cur, kernel)[:, :, ::2, ::2]
        pyr.append(cur - upsample(down))
        cur = down
    return pyr


def lap_loss(inp, target, max_levels=3):
    return sum(F.l1_loss(a, b) for a, b in zip(laplacian_pyramid(inp, max_levels), laplacian_pyramid(target, max_levels)))
