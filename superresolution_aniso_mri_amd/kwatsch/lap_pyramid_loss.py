"""Laplacian-pyramid L1 loss: the reference's ``kwatsch/lap_pyramid_loss.py`` on HIP kernels (csrc/lap.hip).

    pyramid level k:  filtered = G(current);  down = filtered[::2, ::2];  diff_k = current - 4*G(zero_insert(down));  current = down
    LapLoss(input, target) = sum_k mean |diff_k(input) - diff_k(target)|                                  (reference :43-65)

G is the 5x5 binomial filter with reflect padding.  Every piece is an autograd Function whose backward launches the transposed
kernel (``aesr_lap_blur5(adjoint=1)``, ``down2`` <-> ``zero_insert2``), so gradients flow to ``input`` as in the reference.
Levels need even H and W (the reference fails with a shape mismatch otherwise).  No CPU fallback."""
import torch

from .. import _hip
from .._hip import check, lib, ptr, stream


def _planes(img):
    """[N,C,H,W] (any memory layout) or [P,H,W] -> contiguous planes [P,H,W]."""
    if img.dim() == 4:
        img = img.contiguous().reshape(img.shape[0] * img.shape[1], img.shape[2], img.shape[3])
    _hip.require_gpu_tensor(img, "LapLoss input")
    return img.contiguous().float()


class _BlurFn(torch.autograd.Function):
    """out = (add or 0) + gain * G(x)."""

    @staticmethod
    def forward(ctx, x, gain, add):
        P, H, W = x.shape
        out = torch.empty_like(x)
        check(lib.aesr_lap_blur5(ptr(x), ptr(add), ptr(out), P, H, W, float(gain), 0, stream()), "aesr_lap_blur5")
        ctx.gain = float(gain)
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        P, H, W = g.shape
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(g)
            check(lib.aesr_lap_blur5(ptr(g), None, ptr(dx), P, H, W, ctx.gain, 1, stream()), "aesr_lap_blur5(adjoint)")
        return dx, None, (g if ctx.needs_input_grad[2] else None)


class _Down2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        P, H, W = x.shape
        out = torch.empty((P, (H + 1) // 2, (W + 1) // 2), device=x.device, dtype=torch.float32)
        check(lib.aesr_lap_down2(ptr(x), ptr(out), P, H, W, stream()), "aesr_lap_down2")
        ctx.hw = (H, W)
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        P, h, w = g.shape
        H, W = ctx.hw
        dx = torch.empty((P, H, W), device=g.device, dtype=torch.float32)
        check(lib.aesr_lap_zero_insert2(ptr(g), ptr(dx), P, h, w, H, W, stream()), "aesr_lap_zero_insert2")
        return dx


class _ZeroInsert2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        P, h, w = x.shape
        out = torch.empty((P, 2 * h, 2 * w), device=x.device, dtype=torch.float32)
        check(lib.aesr_lap_zero_insert2(ptr(x), ptr(out), P, h, w, 2 * h, 2 * w, stream()), "aesr_lap_zero_insert2")
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        P, H, W = g.shape
        dx = torch.empty((P, H // 2, W // 2), device=g.device, dtype=torch.float32)
        check(lib.aesr_lap_down2(ptr(g), ptr(dx), P, H, W, stream()), "aesr_lap_down2")
        return dx


class _L1Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        partial = torch.empty(_hip.MSE_NPART, device=a.device, dtype=torch.float64)
        loss = torch.empty(1, device=a.device, dtype=torch.float32)
        check(lib.aesr_l1_fwd(ptr(a), ptr(b), ptr(partial), ptr(loss), a.numel(), stream()), "aesr_l1_fwd")
        ctx.save_for_backward(a, b)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.reshape(1).contiguous().float()
        da = torch.empty_like(a)
        check(lib.aesr_l1_bwd(ptr(a), ptr(b), ptr(g), ptr(da), a.numel(), stream()), "aesr_l1_bwd")
        return (da if ctx.needs_input_grad[0] else None), (-da if ctx.needs_input_grad[1] else None)


def conv_gauss(planes, gain=1.0, add=None):
    return _BlurFn.apply(planes, gain, add)


def downsample(planes):
    return _Down2Fn.apply(planes)


def upsample(planes):
    """zero-insert to twice the size, then 4*G (reference :27-34)."""
    return conv_gauss(_ZeroInsert2Fn.apply(planes), 4.0)


def laplacian_pyramid(img, max_levels=3):
    """List of ``max_levels`` difference planes [P, H/2^k, W/2^k] (reference :43-53)."""
    cur = _planes(img)
    pyr = []
    for _ in range(max_levels):
        P, H, W = cur.shape
        if H % 2 or W % 2:
            raise ValueError("LapLoss: every pyramid level needs even height and width, got %dx%d" % (H, W))
        down = downsample(conv_gauss(cur))
        pyr.append(conv_gauss(_ZeroInsert2Fn.apply(down), -4.0, cur))       # current - upsample(down), one launch
        cur = down
    return pyr


class LapLoss(torch.nn.Module):
    """``LapLoss(max_levels=3, channels=3, device=...)`` of the reference (:56-65); ``channels`` only sized the reference's filter
    bank (the same binomial kernel per channel) and is accepted for signature compatibility."""

    def __init__(self, max_levels=3, channels=3, device=None):
        super().__init__()
        self.max_levels = max_levels
        self.channels = channels

    def forward(self, input, target):
        if tuple(input.shape) != tuple(target.shape):
            raise ValueError("LapLoss: shape mismatch %s vs %s" % (tuple(input.shape), tuple(target.shape)))
        pyr_in = laplacian_pyramid(input, self.max_levels)
        with torch.no_grad() if not target.requires_grad else torch.enable_grad():
            pyr_tg = laplacian_pyramid(target, self.max_levels)
        total = None
        for a, b in zip(pyr_in, pyr_tg):
            term = _L1Fn.apply(a, b)
            total = term if total is None else total + term
        return total
