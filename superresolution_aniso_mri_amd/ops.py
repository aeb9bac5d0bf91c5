"""Autograd wrappers of the non-network HIP ops of the ae_combined step (latent lerp, MSE) and the fused Adam.

Follows kwatsch/cardiac/trainer_ae.py:173 / kwatsch/brain/trainer_ae.py:264-266 (lerp), kwatsch/base_trainer.py:177
(``F.mse_loss`` mean) and kwatsch/trainer_ae.py:29-30 (``optim.Adam``).  No CPU fallback."""
import os
from ctypes import c_float

import numpy as np
import torch

from . import _hip, engine
from .engine import _pb, _pe
from ._hip import check, lib, ptr, stream


def _flat_pair(a, b):
    """Two same-shape tensors -> contiguous buffers with identical element order (NHWC if 4-D)."""
    if a.shape != b.shape:
        raise ValueError("shape mismatch %s vs %s" % (tuple(a.shape), tuple(b.shape)))
    if a.dim() == 4:
        return engine.to_nhwc(a), engine.to_nhwc(b)
    return a.contiguous().float(), b.contiguous().float()


class _LerpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, a_from, a_to):
        B = z.shape[0] // 2
        per = z[0].numel()
        zmix = torch.empty((B,) + tuple(z.shape[1:]), device=z.device, dtype=torch.float32)
        check(lib.aesr_lerp_fwd(ptr(z), ptr(a_from), ptr(a_to), ptr(zmix), B, per, stream()), "aesr_lerp_fwd")
        ctx.save_for_backward(a_from, a_to)
        return zmix

    @staticmethod
    def backward(ctx, dzmix):
        a_from, a_to = ctx.saved_tensors
        dzmix = dzmix.contiguous()
        B = dzmix.shape[0]
        dz = torch.empty((2 * B,) + tuple(dzmix.shape[1:]), device=dzmix.device, dtype=torch.float32)
        check(lib.aesr_lerp_bwd(ptr(dzmix), ptr(a_from), ptr(a_to), ptr(dz), B, dzmix[0].numel(), stream()), "aesr_lerp_bwd")
        return dz, None, None


class _LerpCatFn(torch.autograd.Function):
    """zcat = [z | a_from*z[:B] + a_to*z[B:]] in one kernel; backward folds the mix gradient back onto both halves."""

    @staticmethod
    def forward(ctx, z, a_from, a_to):
        B = z.shape[0] // 2
        zcat = torch.empty((3 * B,) + tuple(z.shape[1:]), device=z.device, dtype=torch.float32)
        _pb("lerp_cat", 0.0, 4.0 * (z.numel() + zcat.numel()))
        check(lib.aesr_lerp_cat_fwd(ptr(z), ptr(a_from), ptr(a_to), ptr(zcat), B, z[0].numel(), stream()), "aesr_lerp_cat_fwd")
        _pe()
        ctx.save_for_backward(a_from, a_to)
        return zcat

    @staticmethod
    def backward(ctx, g):
        a_from, a_to = ctx.saved_tensors
        g = g.contiguous()
        B = g.shape[0] // 3
        dz = torch.empty((2 * B,) + tuple(g.shape[1:]), device=g.device, dtype=torch.float32)
        _pb("lerp_cat", 0.0, 4.0 * (g.numel() + dz.numel()))
        check(lib.aesr_lerp_cat_bwd(ptr(g), ptr(a_from), ptr(a_to), ptr(dz), B, g[0].numel(), stream()), "aesr_lerp_cat_bwd")
        _pe()
        return dz, None, None


_CONST_VECS = {}


def _const_vec(n, value, device):
    """[n] fp32 tensor filled with ``value``, cached per (device, n, value): the scalar mixing coefficients (0.5 / 0.5) are the same
    every step, so no fill kernel runs per call.  While a stream is capturing, a fresh tensor is made instead (a tensor born in a
    graph's private pool must not outlive the graph through this cache)."""
    key = (str(device), int(n), float(value))
    t = _CONST_VECS.get(key)
    if torch.cuda.is_current_stream_capturing():
        # a vector cached by an earlier (eager) step lives outside the graph's pool: safe to bake into the graph, and no fill node
        return t if t is not None else torch.full((n,), float(value), dtype=torch.float32, device=device)
    if t is None:
        if len(_CONST_VECS) > 64:
            _CONST_VECS.clear()
        t = _CONST_VECS[key] = torch.full((n,), float(value), dtype=torch.float32, device=device)
    return t


def _lerp_args(z, alpha_from, alpha_to):
    B = z.shape[0] // 2
    if z.shape[0] != 2 * B or B == 0:
        raise ValueError("lerp_mix needs an even, non-empty batch (got %d)" % z.shape[0])
    zn = engine.to_nhwc(z) if z.dim() == 4 else z.contiguous()
    _hip.require_gpu_tensor(zn, "z")
    if zn[0].numel() % 4 != 0:
        raise ValueError("latent size per image must be a multiple of 4")

    def coef(a):
        if isinstance(a, (int, float)):
            return _const_vec(B, a, z.device)
        a = torch.as_tensor(a, dtype=torch.float32, device=z.device).reshape(-1)
        if a.numel() == 1:
            a = a.expand(B)
        if a.numel() != B:
            raise ValueError("need one mixing coefficient per pair (%d), got %d" % (B, a.numel()))
        return a.contiguous()
    return zn, coef(alpha_from), coef(alpha_to)


def lerp_mix(z, alpha_from, alpha_to):
    """z: logical NCHW [2B,C,H,W] (rows i and i+B are a pair).  alpha_*: [B] or [B,1] or scalar.
    Returns z_mix [B,C,H,W] = alpha_from*z[:B] + alpha_to*z[B:]."""
    zn, af, at = _lerp_args(z, alpha_from, alpha_to)
    out = _LerpFn.apply(zn, af, at)
    return engine.to_nchw_view(out) if z.dim() == 4 else out


def lerp_multi(z_nhwc, alphas, act=_hip.ACT_NONE, slope=0.0):
    """All mixes of neighbouring slices in ONE launch: z_nhwc [Z, ...] -> [n * (Z - 1), ...], row k * (Z - 1) + i =
    act(alphas[k] * z[i + 1] + (1 - alphas[k]) * z[i]) (no gradient: inference)."""
    _hip.require_gpu_tensor(z_nhwc, "z")
    Z, n = z_nhwc.shape[0], len(alphas)
    per = z_nhwc[0].numel()
    out = torch.empty((n * (Z - 1),) + tuple(z_nhwc.shape[1:]), device=z_nhwc.device, dtype=torch.float32)
    for k0 in range(0, n, 16):
        part = [float(a) for a in alphas[k0:k0 + 16]]
        check(lib.aesr_lerp_multi(ptr(z_nhwc), ptr(out[k0 * (Z - 1):]), Z, per, _hip.float_array(part), len(part), int(act), float(slope),
                                  stream()), "aesr_lerp_multi")
    return out


def interleave_clamp(orig, synth, n, lo=0.0, hi=1.0):
    """The super-resolved volume in one pass: orig [Z, H, W], synth [n * (Z - 1), H, W] (row k * (Z - 1) + i, as ``lerp_multi`` orders them)
    -> [(Z - 1)(n + 1) + 1, H, W] with slice i at slot i (n + 1) and its n synthesised successors behind it, clamped to [lo, hi]."""
    _hip.require_gpu_tensor(orig, "orig")
    Z, per = orig.shape[0], orig[0].numel()
    orig = orig.contiguous()
    if Z > 1 and n > 0:
        _hip.require_gpu_tensor(synth, "synth")
        synth = synth.contiguous()
        if synth.numel() != n * (Z - 1) * per:
            raise ValueError("synth holds %d elements, %d x %d slices of %d expected" % (synth.numel(), n, Z - 1, per))
    else:
        synth, n = None, 0
    out = torch.empty(((Z - 1) * (n + 1) + 1,) + tuple(orig.shape[1:]), device=orig.device, dtype=torch.float32)
    check(lib.aesr_interleave_clamp(ptr(orig), ptr(synth), ptr(out), Z, n, per, float(lo), float(hi), stream()), "aesr_interleave_clamp")
    return out


def lerp_cat(z, alpha_from, alpha_to):
    """[z | lerp_mix(z)] as ONE [3B,...] tensor written by one kernel: the decoder input of the ae_combined step (its rows 2B.. are
    z_mix).  Saves the separate concatenation pass and, in backward, the accumulation of the two gradient paths into z."""
    zn, af, at = _lerp_args(z, alpha_from, alpha_to)
    out = _LerpCatFn.apply(zn, af, at)
    return engine.to_nchw_view(out) if z.dim() == 4 else out


class _MseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        partial = torch.empty(_hip.MSE_NPART, device=a.device, dtype=torch.float64)
        loss = torch.empty(1, device=a.device, dtype=torch.float32)
        _pb("mse", 0.0, 8.0 * a.numel())
        check(lib.aesr_mse_fwd(ptr(a), ptr(b), ptr(partial), ptr(loss), a.numel(), stream()), "aesr_mse_fwd")
        _pe()
        ctx.save_for_backward(a, b)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.reshape(1).contiguous().float()
        need_a, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if need_b and not need_a:            # d/db = 2(b-a)g/n: the same kernel with the operands swapped, no negation pass
            db = torch.empty_like(b)
            check(lib.aesr_mse_bwd(ptr(b), ptr(a), ptr(g), ptr(db), b.numel(), stream()), "aesr_mse_bwd")
            return None, db
        da = torch.empty_like(a)
        check(lib.aesr_mse_bwd(ptr(a), ptr(b), ptr(g), ptr(da), a.numel(), stream()), "aesr_mse_bwd")
        return (da if need_a else None), (-da if need_b else None)


def mse_loss(a, b):
    """mean((a-b)^2) over all elements as a 0-dim device tensor (no host sync)."""
    an, bn = _flat_pair(a, b)
    _hip.require_gpu_tensor(an, "mse input")
    _hip.require_gpu_tensor(bn, "mse target")
    return _MseFn.apply(an, bn)


_MSE3_WS = {}


def _mse3_workspace(device, owner=None):
    """Partial sums + ticket counter of aesr_mse3_fwd: zeroed ONCE (the kernel leaves it consistent).  One per (owner, device, stream)
    for eager steps and one per (owner, device) for that owner's captured graphs -- ``owner`` is the trainer (its ``__dict__`` holds
    them), so two trainers whose captured steps replay on different streams never share partial sums and the ticket; callers
    without an owner share the module-level set.  Allocated outside any graph capture (the first, eager steps)."""
    store = owner.__dict__.setdefault("_aesr_mse3_ws", {}) if owner is not None else _MSE3_WS
    if torch.cuda.is_current_stream_capturing():
        # a capture runs on a stream of its own: the captured graphs of ONE owner (replayed one after the other on one stream) share
        # one workspace, created by that owner's eager steps before the capture -- a fresh one here would be a memset node in every replay
        key = (str(device), "graph")
        if key not in store:
            raise RuntimeError("aesr_mse3_fwd: run the loss eagerly once before capturing it into a HIP graph")
        return store[key]
    key = (str(device), int(torch.cuda.current_stream(device).cuda_stream))
    if key not in store:
        store[key] = torch.zeros(_hip.MSE3_WS, device=device, dtype=torch.float64)
        store.setdefault((str(device), "graph"), torch.zeros(_hip.MSE3_WS, device=device, dtype=torch.float64))
    return store[key]


class _CombinedMseFn(torch.autograd.Function):
    """total = mse(o3[:n1], x) + lam * mse(o3[n1:], between), plus the logged mse(z_mix, z_ref), in ONE launch; the backward writes
    the whole gradient of o3 in one launch (no split / cat of the decoder's two sub-batches)."""

    @staticmethod
    def forward(ctx, o3, x, between, z_mix, z_ref, lam, owner=None):
        n1, n2 = x.numel(), between.numel()
        if o3.numel() != n1 + n2:
            raise ValueError("combined_mse: %d outputs for %d + %d targets" % (o3.numel(), n1, n2))
        ctx.set_materialize_grads(False)       # the three logged outputs carry no gradient: no zero tensors for them
        res = torch.empty(4, device=o3.device, dtype=torch.float32)
        flat = o3.reshape(-1)
        _pb("mse3", 0.0, 4.0 * (2 * (n1 + n2) + 2 * (z_mix.numel() if z_mix is not None else 0)))
        check(lib.aesr_mse3_fwd(ptr(flat), ptr(x), n1, ptr(flat[n1:]), ptr(between), n2, ptr(z_mix), ptr(z_ref),
                                z_mix.numel() if z_mix is not None else 0, ptr(lam), ptr(_mse3_workspace(o3.device, owner)), ptr(res), stream()),
              "aesr_mse3_fwd")
        _pe()
        ctx.save_for_backward(o3, x, between, lam)
        outs = tuple(res[i].reshape(()) for i in range(4))
        ctx.mark_non_differentiable(*outs[1:])
        return outs

    @staticmethod
    def backward(ctx, g, *_unused):
        if g is None:
            return None, None, None, None, None, None, None
        o3, x, between, lam = ctx.saved_tensors
        n1, n2 = x.numel(), between.numel()
        g = g.reshape(1).contiguous().float()
        d = torch.empty_like(o3)
        flat, dflat = o3.reshape(-1), d.reshape(-1)
        _pb("mse3", 0.0, 4.0 * 3 * (n1 + n2))
        check(lib.aesr_mse3_bwd(ptr(flat), ptr(x), n1, ptr(flat[n1:]), ptr(between), n2, ptr(lam), ptr(g), ptr(dflat), ptr(dflat[n1:]),
                                stream()), "aesr_mse3_bwd")
        _pe()
        return d, None, None, None, None, None, None


def combined_mse(o3, x, between, z_mix, z_ref, lam, owner=None):
    """The loss block of the ae_combined step with MSE losses (kwatsch/cardiac/trainer_ae.py:160-182 of the reference):
    o3 = the decoder's batched output [recon(x) | synthesized between-slices] (logical NCHW), x / between their targets, lam the
    synthesis weight as a device scalar.  Returns (total, loss_rec, lam * loss_img, loss_latent) as 0-dim device tensors; only
    ``total`` carries a gradient (to o3).  ``owner``: the object whose steps these are (the trainer): keeps the kernel's partial-sum
    workspace apart from other owners' (``_mse3_workspace``)."""
    o3n = engine.to_nhwc(o3)
    xn, bn = engine.to_nhwc(x), engine.to_nhwc(between)
    zm, zr = _flat_pair(z_mix, z_ref)
    for t, what in ((o3n, "decoder output"), (xn, "image"), (bn, "slice_between"), (zm, "z_mix"), (zr, "z_ref"), (lam, "lambda")):
        _hip.require_gpu_tensor(t, "combined_mse " + what)
    total, l_rec, l_img, l_lat = _CombinedMseFn.apply(o3n, xn, bn, zm, zr, lam.reshape(1), owner)
    return total, l_rec, l_img, l_lat


class _RowMeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        N, M = x.shape[0], x[0].numel()
        out = torch.empty(N, device=x.device, dtype=torch.float32)
        check(lib.aesr_row_mean_fwd(ptr(x), ptr(out), N, M, stream()), "aesr_row_mean_fwd")
        ctx.shape = tuple(x.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().float()
        dx = torch.empty(ctx.shape, device=g.device, dtype=torch.float32)
        check(lib.aesr_row_mean_bwd(ptr(g), ptr(dx), ctx.shape[0], dx[0].numel(), stream()), "aesr_row_mean_bwd")
        return dx


def row_mean(x):
    """Mean over everything but the first dimension, [N, ...] -> [N] (element order is irrelevant, so NHWC memory is used as is)."""
    xn = engine.to_nhwc(x) if x.dim() == 4 else x.contiguous().float()
    _hip.require_gpu_tensor(xn, "row_mean input")
    return _RowMeanFn.apply(xn)


class HipAdam(torch.optim.Adam):
    """``torch.optim.Adam`` whose step is ONE fused HIP kernel over a flat parameter buffer.

    Parameters are re-pointed at views of one flat fp32 buffer (same Parameter objects, so module state_dicts are
    untouched); ``p.grad`` are views of a flat gradient buffer and ``exp_avg`` / ``exp_avg_sq`` views of flat moment
    buffers, so ``state_dict()`` keeps the stock Adam layout the reference checkpoints use
    (kwatsch/base_trainer.py:353-362).  The step counter lives on the device (graph-replay safe)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, on_step=None):
        params = [p for p in params]
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self.on_step = on_step
        self._plist = [p for g in self.param_groups for p in g["params"]]
        if len(self.param_groups) != 1:
            raise NotImplementedError("HipAdam supports a single parameter group")
        dev = self._plist[0].device
        if dev.type != "cuda":
            raise RuntimeError("HipAdam needs parameters on the GPU (got %s): no CPU fallback" % dev)
        n = sum(p.numel() for p in self._plist)
        self.flat_p = torch.empty(n, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(n, device=dev, dtype=torch.float32)
        self.flat_m = torch.zeros(n, device=dev, dtype=torch.float32)
        self.flat_v = torch.zeros(n, device=dev, dtype=torch.float32)
        self.dev_state = torch.zeros(8, device=dev, dtype=torch.float32)
        self._init_state(0.0)
        self._host_step = 0
        self.lazy_zero = os.environ.get("AESR_LAZY_ZERO", "1") != "0"
        self._pending_zero = False
        off = 0
        for p in self._plist:
            k = p.numel()
            self.flat_p[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat_p[off:off + k].view_as(p.data)
            p.grad = self.flat_g[off:off + k].view_as(p.data)
            self.state[p] = {"step": torch.tensor(0.0), "exp_avg": self.flat_m[off:off + k].view_as(p.data),
                             "exp_avg_sq": self.flat_v[off:off + k].view_as(p.data)}
            off += k
        self.numel = n

    def _init_state(self, steps_done):
        """Device state of aesr_adam_step for an optimizer that has taken ``steps_done`` steps (bias corrections of the next one)."""
        b1, b2 = self.param_groups[0]["betas"]
        host = (c_float * 8)()
        lib.aesr_adam_state_init(host, float(steps_done), float(b1), float(b2))
        # raw bytes: floats 4..7 carry two doubles, whose halves need not be meaningful floats
        self.dev_state.view(torch.int32).copy_(torch.from_numpy(np.frombuffer(host, dtype=np.int32).copy()))
        self._state_betas = (float(b1), float(b2))

    def _check_views(self):
        off = 0
        for p in self._plist:
            k = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * off:
                g_old = p.grad
                p._aesr_grad_fresh = False
                p.grad = self.flat_g[off:off + k].view_as(p.data)
                if g_old is not None:
                    p.grad.copy_(g_old)
                else:
                    p.grad.zero_()
            off += k

    def zero_grad(self, set_to_none=False):
        """Gradients count as zero from here on.  ``lazy_zero`` (default): no memset node -- the engine OVERWRITES the gradient of
        every parameter it differentiates (``_aesr_grad_fresh``), and ``step()`` zeroes the few it did not reach before Adam reads
        them; until then the buffer of an unreached parameter still holds the previous step's values."""
        self._check_views()
        if not self.lazy_zero:
            self.flat_g.zero_()
        for p in self._plist:
            p._aesr_grad_fresh = True      # engine may overwrite p.grad directly instead of going through autograd's +=
        self._pending_zero = self.lazy_zero

    def _settle_unwritten(self):
        """lazy_zero: parameters no backward pass wrote since zero_grad() get their zeros now."""
        if self._pending_zero:
            for p in self._plist:
                if getattr(p, "_aesr_grad_fresh", False):
                    p.grad.zero_()
            self._pending_zero = False

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("closures are not supported")
        self._check_views()
        self._settle_unwritten()
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        if (float(b1), float(b2)) != self._state_betas:            # betas changed between steps: the running powers start over
            self._init_state(float(self.dev_state[0].item()))
        _pb("adam_step", 0.0, 28.0 * self.numel)
        check(lib.aesr_adam_step(ptr(self.flat_p), ptr(self.flat_g), ptr(self.flat_m), ptr(self.flat_v), ptr(self.dev_state),
                                 self.numel, float(g["lr"]), float(b1), float(b2), float(g["eps"]), float(g["weight_decay"]),
                                 0, stream()), "aesr_adam_step")
        _pe()
        self._host_step += 1
        if self.on_step is not None:
            self.on_step()

    def state_dict(self):
        step = float(self.dev_state[0].item())
        for p in self._plist:
            self.state[p]["step"] = torch.tensor(step)
        return super().state_dict()

    def load_state_dict(self, sd):
        super().load_state_dict(sd)
        off, step = 0, 0.0
        for p in self._plist:
            k = p.numel()
            st = self.state[p]
            if "exp_avg" in st:
                self.flat_m[off:off + k].copy_(st["exp_avg"].reshape(-1))
                self.flat_v[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
                step = float(st["step"])
            st["exp_avg"] = self.flat_m[off:off + k].view_as(p.data)
            st["exp_avg_sq"] = self.flat_v[off:off + k].view_as(p.data)
            off += k
        self._init_state(step)
        self._host_step = int(step)
