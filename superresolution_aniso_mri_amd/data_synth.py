"""Synthetic (from, to, between) slice triplets in the reference's batch layout (``image`` = [2B,1,H,W] with all
*from* slices then all *to* slices, ``slice_between`` = [B,1,H,W]; datasets/ACDC/data4d_simple.py:327-373) for benchmarks,
``--synthetic`` training and smoke tests.  Smooth, MRI-like and correlated across the triplet so lerp / LPIPS are
non-degenerate: sum of Gaussian blobs + low-pass noise; between = 0.5(from+to) + N(0, 0.02)."""
import torch
import torch.nn.functional as F


def synthetic_batch(B, H, W, seed, brain=False):
    """CPU float32 batch dict; deterministic in (B, H, W, seed)."""
    g = torch.Generator().manual_seed(int(seed))
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
    nb = 8
    ctr = torch.rand(B, nb, 2, generator=g)
    sig = 0.04 + 0.12 * torch.rand(B, nb, generator=g)
    amp = 0.2 + 0.5 * torch.rand(B, nb, generator=g)
    shift = 0.03 * torch.randn(B, nb, 2, generator=g)

    def render(c):
        d2 = (yy[None, None] - c[..., 0, None, None]) ** 2 + (xx[None, None] - c[..., 1, None, None]) ** 2
        return (amp[..., None, None] * torch.exp(-d2 / (2 * sig[..., None, None] ** 2))).sum(1)

    def smooth_noise():
        n = torch.randn(B, 1, H, W, generator=g)
        return F.avg_pool2d(F.pad(n, (2, 2, 2, 2), mode="reflect"), 5, stride=1)[:, 0]

    frm = (render(ctr) + 0.05 * smooth_noise()).clamp(0, 1)
    to = (render(ctr + shift) + 0.05 * smooth_noise()).clamp(0, 1)
    batch = {}
    if brain:
        # datasets/common_brains.py:117-119: coefficients from the slice distances, alpha_to = 1 - alpha_from
        af = torch.tensor([0.25, 0.5, 0.75])[torch.randint(0, 3, (B,), generator=g)][:, None]
        batch["alpha_from"], batch["alpha_to"] = af, 1 - af
        mid = af[:, :, None] * frm + (1 - af[:, :, None]) * to
    else:
        mid = 0.5 * (frm + to)
    between = (mid + 0.02 * torch.randn(B, H, W, generator=g)).clamp(0, 1)
    batch["image"] = torch.cat([frm[:, None], to[:, None]], dim=0).float().contiguous()
    batch["slice_between"] = between[:, None].float().contiguous()
    batch["loss_mask"] = torch.ones(2 * B, 1, H, W)
    return batch


def shard_batch(batch, rank, world):
    """Data parallel: rank r keeps triplets [lo, hi) and rebuilds ``image`` so row i still pairs with row i + B_local."""
    B = batch["slice_between"].shape[0]
    lo, hi = (B * rank) // world, (B * (rank + 1)) // world
    out = {}
    for k, v in batch.items():
        if k == "_persistent":
            continue              # the shard is made of fresh tensors: not the producer's persistent output buffer
        if k in ("image", "loss_mask") and v.shape[0] == 2 * B:
            out[k] = torch.cat([v[lo:hi], v[B + lo:B + hi]], dim=0)
        elif torch.is_tensor(v) and v.shape[0] == B:
            out[k] = v[lo:hi]
        else:
            out[k] = v
    return out
