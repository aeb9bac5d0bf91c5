"""MI355X-native ``VanillaACAI`` auto-encoder: same constructor arguments, state_dict keys and initialisation as
the reference's networks/acai_vanilla.py:39-138, but every forward/backward runs on the HIP kernels of
libaesr_hip.so (no ATen convolution / batch-norm is ever dispatched).

The ``enc`` / ``dec`` attributes are ``nn.Sequential`` containers of stock torch modules used purely as parameter
holders (checkpoint compatibility); ``HipAE`` executes them through
``superresolution_aniso_mri_amd.engine``.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from .. import engine

activation = nn.LeakyReLU


def Initializer(layers, slope=0.2):
    """Reference init (networks/acai_vanilla.py:39-46): N(0, 1/sqrt((1+slope^2)*prod(w.shape[:-1]))) for every
    layer that has a ``weight`` (BatchNorm gamma included), zero biases."""
    for layer in layers:
        if hasattr(layer, "weight"):
            w = layer.weight.data
            w.normal_(std=1 / np.sqrt((1 + slope ** 2) * np.prod(w.shape[:-1])))
        if hasattr(layer, "bias"):
            layer.bias.data.zero_()


def _block(kp, k, use_batchnorm):
    layers = [nn.Conv2d(kp, k, 3, padding=1), activation(), nn.Conv2d(k, k, 3, padding=1), activation()]
    if use_batchnorm:
        layers.append(nn.BatchNorm2d(k))
    return layers


def Encoder(scales, depth, latent, colors, n_res_block=None, use_batchnorm=False, stem=None, downsample="avgpool"):
    """networks/acai_vanilla.py:49-72 (``stem``/``downsample`` cover the LargerAE / strided variants)."""
    if n_res_block is not None:
        raise NotImplementedError("n_res_block is not used by any ae_combined configuration")
    stem = depth if stem is None else stem
    layers = [nn.Conv2d(colors, stem, 1, padding=1)]
    kp = stem
    for scale in range(scales):
        k = depth << scale
        layers += _block(kp, k, use_batchnorm)
        layers.append(nn.AvgPool2d(2) if downsample == "avgpool" else nn.Conv2d(k, k, 2, stride=2, padding=0))
        kp = k
    k = depth << scales
    layers += [nn.Conv2d(kp, k, 3, padding=1), activation(), nn.Conv2d(k, latent, 3, padding=1)]
    Initializer(layers)
    return nn.Sequential(*layers)


def Decoder(scales, depth, latent, colors, n_res_block=None, use_upsample=True, use_batchnorm=False, use_sigmoid=False,
            stem_1x1=False, upsample_mode="nearest"):
    """networks/acai_vanilla.py:75-102 (``stem_1x1``: the LargerAE 1x1 conv (+BN) in front; ``upsample_mode="bilinear"``:
    the align_corners=False bilinear x2 upsample of networks/ae_standard.py:68 instead of nearest)."""
    if n_res_block is not None or not use_upsample:
        raise NotImplementedError("ResBlock / ConvTranspose decoders are not part of the ae_combined path")
    layers, kp = [], latent
    if stem_1x1:
        c1 = depth << scales
        layers += [nn.Conv2d(latent, c1, 1, padding=0), activation()]
        if use_batchnorm:
            layers.append(nn.BatchNorm2d(c1))
        kp = c1
    for scale in range(scales - 1, -1, -1):
        k = depth << scale
        layers += _block(kp, k, use_batchnorm)
        layers.append(nn.Upsample(scale_factor=2) if upsample_mode == "nearest" else
                      nn.Upsample(scale_factor=2, mode=upsample_mode, align_corners=False))
        kp = k
    layers += [nn.Conv2d(kp, depth, 3, padding=1), activation()]
    layers.append(nn.Conv2d(depth, colors, 3, padding=1))
    if use_sigmoid:
        layers.append(nn.Sigmoid())
    Initializer(layers)
    return nn.Sequential(*layers)


def _joined_view(xs):
    """The sub-batches as ONE [sum N, H, W, C] tensor WITHOUT a copy when they already lie back to back in one allocation (the
    static input buffer of the captured step, kwatsch/trainer_ae.py), else None."""
    first = xs[0]
    if any(t.shape[1:] != first.shape[1:] or t.dtype != first.dtype or not t.is_contiguous() for t in xs):
        return None
    end = first.data_ptr() + first.numel() * first.element_size()
    for t in xs[1:]:
        if t.data_ptr() != end or t.untyped_storage().data_ptr() != first.untyped_storage().data_ptr():
            return None
        end += t.numel() * t.element_size()
    n = sum(t.shape[0] for t in xs)
    return first.new_empty(0).set_(first.untyped_storage(), first.storage_offset(), (n,) + tuple(first.shape[1:]))


def num_scales(args):
    return int(round(math.log(args["width"] // args["latent_width"], 2)))


class HipAE(nn.Module):
    """Common behaviour of the auto-encoder family: encode / decode / forward on the HIP engine."""

    def _fill_defaults(self, args):
        args.setdefault("n_res_block", None)
        args.setdefault("use_batchnorm", False)
        args.setdefault("use_sigmoid", False)
        args.setdefault("gpu_ids", [0])

    def _runner(self, name):
        cache = self.__dict__.setdefault("_runners", {})
        r = cache.get(name)
        if r is None:
            r = engine.SequentialRunner(getattr(self, name))
            cache[name] = r
        return r

    def mark_weights_dirty(self):
        for r in self.__dict__.get("_runners", {}).values():
            r.mark_weights_dirty()

    def prepare_weights(self, names=("enc", "dec")):
        """Ready the parameter-side operands of ALL networks of a training step with one launch (engine.prepare_weights) instead of one
        to three per network when their passes start; a no-op when nothing changed since the last call."""
        engine.prepare_weights([(self._runner(n), self._runner(n).train_steps()) for n in names if hasattr(self, n)])

    def ensure_bn_barriers(self, names=("enc", "dec")):
        """Create the grid-barrier state of the one-launch BatchNorm kernels now (outside any graph capture): a step that is captured at a
        batch size the eager steps never saw would otherwise meet ``SequentialRunner._bn_barrier`` for the first time inside the capture."""
        for n in names:
            if hasattr(self, n):
                p = next(getattr(self, n).parameters(), None)
                if p is not None and p.is_cuda:
                    self._runner(n)._bn_barrier(p.device)

    def set_sync_bn(self, fn, count_scale=1.0, p2p=None):
        """Data parallel: ``fn(sums)`` all-reduces BatchNorm partial sums across ranks (SyncBN); ``count_scale`` =
        B_global / B_local turns local element counts into global ones.  ``p2p``: a parallel.PeerExchange -- the exchange then happens
        inside the one-launch BatchNorm kernels wherever the layer fits them (``fn`` stays the way for the others)."""
        for name in ("enc", "dec"):
            r = self._runner(name)
            r.sync_bn, r.count_scale, r.sync_p2p = fn, float(count_scale), p2p

    # -- multi-group passes (one launch sequence, independent BatchNorm statistics per sub-batch) ---------
    def _pass(self, name, tensors, needs_grad=None):
        if needs_grad is None:
            needs_grad = [True] * len(tensors)
        if any(g and not p for p, g in zip(needs_grad[:-1], needs_grad[1:])):
            raise ValueError("sub-batches that need gradients must come first")
        xs = [engine.to_nhwc(t) for t in tensors]
        x = xs[0] if len(xs) == 1 else (_joined_view(xs) if not any(t.requires_grad for t in xs) else None)
        if x is None:
            x = torch.cat(xs, dim=0)
        return self._pass_batched(name, x, [t.shape[0] for t in xs], needs_grad)

    def _pass_batched(self, name, x_nhwc, splits, needs_grad, merge=False):
        nstart = [0]
        for n in splits:
            nstart.append(nstart[-1] + int(n))
        ngrad = sum(int(n) for n, g in zip(splits, needs_grad) if g)
        if not self.training:
            nstart = [0, nstart[-1]]          # eval: running statistics, groups are irrelevant
            ngrad = nstart[-1] if ngrad > 0 else 0
        outs = [sum(int(n) for n in splits)] if merge else list(splits)     # merge: ONE output tensor, the statistic groups stay
        return engine.run_pass_groups(self._runner(name), x_nhwc, outs, nstart, ngrad, train=self.training)

    def decode_cat(self, zcat, splits, needs_grad=None, merge=False):
        """``decode_multi`` on sub-batches that already sit in one tensor (``ops.lerp_cat``): no concatenation pass.  ``merge``
        returns the outputs of all sub-batches as ONE tensor (a list of one), still with per-sub-batch BatchNorm statistics."""
        needs_grad = [True] * len(splits) if needs_grad is None else needs_grad
        if sum(int(n) for n in splits) != zcat.shape[0]:
            raise ValueError("splits %s do not add up to the batch size %d" % (list(splits), zcat.shape[0]))
        return self._pass_batched("dec", engine.to_nhwc(zcat), splits, needs_grad, merge=merge)

    def encode_multi(self, images, needs_grad=None):
        return self._pass("enc", images, needs_grad)

    def decode_multi(self, latents, needs_grad=None):
        return self._pass("dec", latents, needs_grad)

    def decode_mixes(self, z, alphas):
        """Eval-mode slice synthesis for a whole volume: z [Z, C, h, w] latents of consecutive slices -> decoded mixes
        [n * (Z - 1), colors, H, W], row k * (Z - 1) + i = dec(alphas[k] * z[i + 1] + (1 - alphas[k]) * z[i]).  The decoder's
        first convolution is linear, so it runs ONCE PER SLICE on z; the mixes are formed on its pre-activations (one launch for all
        of them, its LeakyReLU applied there) and only the rest of the decoder runs per synthesised slice.  Returns None when the
        decoder does not start with conv + LeakyReLU/ReLU/nothing (caller falls back to lerp on the latents + a full decode)."""
        from .. import _hip, ops
        if self.training:
            raise RuntimeError("decode_mixes is an inference path (model.eval())")
        r = self._runner("dec")
        s0 = r.steps[0]
        if s0.kind != "conv" or s0.s2d or s0.act not in (_hip.ACT_NONE, _hip.ACT_LRELU, _hip.ACT_RELU) or len(r.steps) < 2 or \
                (s0.act == _hip.ACT_LRELU and not 0.0 <= s0.slope <= 1.0):
            return None
        zn = engine.to_nhwc(z)
        N = zn.shape[0]
        with torch.no_grad():
            pre, _, _ = r.forward(zn, (0, N), False, save=False, first=0, last=1, raw_last=True)
            mixed = ops.lerp_multi(pre, alphas, s0.act, s0.slope)
            # big volumes in pieces: no activation tensor of a pass may exceed the kernels' 32-bit offset range -- 2^28 elements (1 GB) per
            # tensor, against the largest one the rest of the decoder really makes (a dHCP volume of 30 slices x 3 mixes goes in ONE pass;
            # the 64-channels-at-full-size guess of BaseTrainer._run_eval cut it into 83 + 4 images: seven launches for the 4)
            n_max = max(1, (1 << 28) // r.max_elems_per_image(mixed.shape[1], mixed.shape[2], mixed.shape[3], first=1))
            outs = [r.forward(mixed[i:i + n_max], (0, min(n_max, mixed.shape[0] - i)), False, save=False, first=1)[0]
                    for i in range(0, mixed.shape[0], n_max)]
            out = outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)
        return engine.to_nchw_view(out)

    def max_elems_per_image(self, what, chw):
        """Largest activation tensor (elements per image) of ``encode`` / ``decode`` / ``forward`` for one C x H x W input image, or None
        when this network is not an ``enc`` / ``dec`` pair of compiled stacks (BaseTrainer._run_eval then keeps its conservative guess)."""
        if not (hasattr(self, "enc") and hasattr(self, "dec")):
            return None
        C, H, W = (int(v) for v in chw)
        if what == "encode":
            return self._runner("enc").trace_shapes(H, W, C)[0]
        if what == "decode":
            return self._runner("dec").trace_shapes(H, W, C)[0]
        if what == "forward":
            e, (h, w, c) = self._runner("enc").trace_shapes(H, W, C)
            return max(e, self._runner("dec").trace_shapes(h, w, c)[0])
        return None

    def encode(self, img):
        return self._pass("enc", [img])[0]

    def decode(self, z):
        return self._pass("dec", [z])[0]

    def forward(self, img):
        return self.decode(self.encode(img))

    def load_state_dict(self, *a, **kw):
        r = super().load_state_dict(*a, **kw)
        self.mark_weights_dirty()
        return r


class VanillaACAI(HipAE):

    def __init__(self, args):
        super().__init__()
        scales = num_scales(args)
        self._fill_defaults(args)
        self.enc = Encoder(scales, args["depth"], args["latent"], args["colors"], n_res_block=args["n_res_block"],
                           use_batchnorm=args["use_batchnorm"]).to(args["device"])
        self.dec = Decoder(scales, args["depth"], args["latent"], args["colors"], n_res_block=args["n_res_block"],
                           use_batchnorm=args["use_batchnorm"], use_sigmoid=args["use_sigmoid"],
                           upsample_mode=args.get("upsample_mode", "nearest")).to(args["device"])


class Discriminator(HipAE):
    """ACAI critic (reference :140-153): an ``Encoder`` of the auto-encoder's shape followed by the mean over all latent elements
    of each sample; ``use_sigmoid`` is hard-wired off as in the reference.  The input may require a gradient (the critic's
    opinion of decoded mixes regularises the auto-encoder), so its first layers run unfolded."""

    def __init__(self, args):
        super().__init__()
        self._fill_defaults(args)
        self.use_sigmoid = False
        self.encoder = Encoder(num_scales(args), args["depth"], args["latent"], args["colors"], n_res_block=args["n_res_block"],
                               use_batchnorm=args["use_batchnorm"]).to(args["device"])

    def set_sync_bn(self, fn, count_scale=1.0, p2p=None):
        r = self._runner("encoder")
        r.sync_bn, r.count_scale, r.sync_p2p = fn, float(count_scale), p2p

    def forward_multi(self, images):
        """One critic value per image for several sub-batches with independent BatchNorm statistics: list of [N_i] tensors."""
        from .. import ops
        return [ops.row_mean(f) for f in self._pass("encoder", images)]

    def forward(self, x):
        return self.forward_multi([x])[0]


def create_decoder(args):
    return Decoder(num_scales(args), args["depth"], args["latent"], args["colors"], n_res_block=args["n_res_block"],
                   use_batchnorm=args["use_batchnorm"], use_sigmoid=args["use_sigmoid"]).to(args["device"])


def lerp(start, end, weights):
    return start + weights * (end - start)


def swap_halves(x):
    a, b = x.split(x.shape[0] // 2)
    return torch.cat([b, a])


def L2(x):
    return torch.mean(x ** 2)
