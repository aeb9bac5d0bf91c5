"""Oracle (TEST INFRASTRUCTURE, see oracle/__init__.py): PyTorch-CPU fp32 restatement of the
reference auto-encoders.

Follows
  * networks/acai_vanilla.py:39-46   (Initializer), :49-72 (Encoder), :75-102 (Decoder),
    :112-138 (VanillaACAI)
  * networks/acai_vanilla_strided.py:9-26  (stride-2 2x2 conv instead of AvgPool)
  * networks/acai_vanilla_modified.py:22-40, :43-68 (depth//2 stem; 1x1 conv (+BN) decoder stem)
  * networks/ae_standard.py:60-80 (only the bilinear x2 upsample it contributes, see `upsample_mode`)

The networks are described as a flat *layer program* whose positions are the indices of the
reference's ``nn.Sequential`` (so parameter names are the reference's state_dict keys,
SURVEY.md App. B) and executed functionally on a plain ``dict`` of tensors.
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

LRELU_SLOPE = 0.01      # nn.LeakyReLU() default, networks/acai_vanilla.py:17
BN_EPS = 1e-5           # nn.BatchNorm2d defaults
BN_MOMENTUM = 0.1


def num_scales(width, latent_width):
    # networks/acai_vanilla.py:116
    return int(round(math.log(width // latent_width, 2)))


def encoder_program(ae_class, scales, depth, latent, colors, use_batchnorm):
    """List of (op, kwargs); list position == nn.Sequential index in the reference."""
    prog = []
    stem = depth // 2 if ae_class == "LargerAE" else depth
    prog.append(("conv", dict(cin=colors, cout=stem, k=1, pad=1, stride=1)))
    kp = stem
    for s in range(scales):
        k = depth << s
        prog += [("conv", dict(cin=kp, cout=k, k=3, pad=1, stride=1)), ("lrelu", {}),
                 ("conv", dict(cin=k, cout=k, k=3, pad=1, stride=1)), ("lrelu", {})]
        if use_batchnorm:
            prog.append(("bn", dict(c=k)))
        if ae_class == "VanillaACAIStrided":
            prog.append(("conv", dict(cin=k, cout=k, k=2, pad=0, stride=2)))
        else:
            prog.append(("avgpool", {}))
        kp = k
    k = depth << scales
    prog += [("conv", dict(cin=kp, cout=k, k=3, pad=1, stride=1)), ("lrelu", {}),
             ("conv", dict(cin=k, cout=latent, k=3, pad=1, stride=1))]
    return prog


def decoder_program(ae_class, scales, depth, latent, colors, use_batchnorm, use_sigmoid,
                    upsample_mode="nearest"):
    prog = []
    kp = latent
    if ae_class == "LargerAE":
        c1 = depth << scales
        prog += [("conv", dict(cin=latent, cout=c1, k=1, pad=0, stride=1)), ("lrelu", {})]
        if use_batchnorm:
            prog.append(("bn", dict(c=c1)))
        kp = c1
    for s in range(scales - 1, -1, -1):
        k = depth << s
        prog += [("conv", dict(cin=kp, cout=k, k=3, pad=1, stride=1)), ("lrelu", {}),
                 ("conv", dict(cin=k, cout=k, k=3, pad=1, stride=1)), ("lrelu", {})]
        if use_batchnorm:
            prog.append(("bn", dict(c=k)))
        prog.append(("upsample", dict(mode=upsample_mode)))
        kp = k
    prog += [("conv", dict(cin=kp, cout=depth, k=3, pad=1, stride=1)), ("lrelu", {}),
             ("conv", dict(cin=depth, cout=colors, k=3, pad=1, stride=1))]
    if use_sigmoid:
        prog.append(("sigmoid", {}))
    return prog


def _as_modules(prog):
    """Stock torch modules in reference construction order (used only for RNG-exact init)."""
    mods = []
    for op, kw in prog:
        if op == "conv":
            mods.append(nn.Conv2d(kw["cin"], kw["cout"], kw["k"], stride=kw["stride"], padding=kw["pad"]))
        elif op == "bn":
            mods.append(nn.BatchNorm2d(kw["c"]))
        elif op == "lrelu":
            mods.append(nn.LeakyReLU())
        elif op == "avgpool":
            mods.append(nn.AvgPool2d(2))
        elif op == "upsample":
            mods.append(nn.Upsample(scale_factor=2))
        elif op == "sigmoid":
            mods.append(nn.Sigmoid())
    return mods


def reference_init_(mods, slope=0.2):
    """networks/acai_vanilla.py:39-46: every layer with a ``.weight`` (conv AND BatchNorm) gets
    N(0, 1/sqrt((1+slope^2) * prod(w.shape[:-1]))); every ``.bias`` is zeroed."""
    for m in mods:
        if hasattr(m, "weight"):
            w = m.weight.data
            std = 1 / np.sqrt((1 + slope ** 2) * np.prod(w.shape[:-1]))
            w.normal_(std=std)
        if hasattr(m, "bias"):
            m.bias.data.zero_()


class OracleAE:
    """Functional AE over a dict of tensors keyed like the reference state_dict."""

    def __init__(self, args, ae_class="VanillaACAI", upsample_mode="nearest", init=True):
        self.ae_class = ae_class
        self.scales = num_scales(args["width"], args["latent_width"])
        bn = bool(args.get("use_batchnorm", False))
        sg = bool(args.get("use_sigmoid", False))
        self.enc_prog = encoder_program(ae_class, self.scales, args["depth"], args["latent"], args["colors"], bn)
        self.dec_prog = decoder_program(ae_class, self.scales, args["depth"], args["latent"], args["colors"],
                                        bn, sg, upsample_mode)
        self.params = OrderedDict()    # learnable, in model.parameters() order
        self.buffers = OrderedDict()   # running_mean / running_var / num_batches_tracked
        if init:
            # same construction + init order as VanillaACAI.__init__ (enc first, then dec), so the
            # same torch seed gives bit-identical parameters to the reference
            for part, prog in (("enc", self.enc_prog), ("dec", self.dec_prog)):
                mods = _as_modules(prog)
                reference_init_(mods)
                for i, m in enumerate(mods):
                    for n, p in m.named_parameters():
                        self.params["%s.%d.%s" % (part, i, n)] = p.detach().clone().requires_grad_(True)
                    for n, b in m.named_buffers():
                        self.buffers["%s.%d.%s" % (part, i, n)] = b.detach().clone()

    # -- state dict plumbing -------------------------------------------------------------
    def state_dict(self):
        sd = OrderedDict()
        for part, prog in (("enc", self.enc_prog), ("dec", self.dec_prog)):
            for i, (op, _) in enumerate(prog):
                names = {"conv": ("weight", "bias"),
                         "bn": ("weight", "bias", "running_mean", "running_var", "num_batches_tracked")}.get(op, ())
                for n in names:
                    k = "%s.%d.%s" % (part, i, n)
                    sd[k] = (self.params[k] if k in self.params else self.buffers[k]).detach().clone()
        return sd

    def load_state_dict(self, sd):
        self.params, self.buffers = OrderedDict(), OrderedDict()
        for k, v in sd.items():
            v = torch.as_tensor(v).detach().clone()
            if k.endswith(("running_mean", "running_var", "num_batches_tracked")):
                self.buffers[k] = v
            else:
                self.params[k] = v.float().requires_grad_(True)
        return self

    def parameters(self):
        return list(self.params.values())

    def zero_grad(self):
        for p in self.params.values():
            p.grad = None

    # -- execution ---------------------------------------------------------------------------
    def _run(self, part, prog, x, train, route=None, tag=""):
        """``route`` (oracle/routing.py, optional): records / forces the LeakyReLU sign decisions under the names ``<tag><part>.<i>``."""
        P, B = self.params, self.buffers
        for i, (op, kw) in enumerate(prog):
            key = "%s.%d." % (part, i)
            if op == "conv":
                x = F.conv2d(x, P[key + "weight"], P[key + "bias"], stride=kw["stride"], padding=kw["pad"])
            elif op == "lrelu":
                x = F.leaky_relu(x, LRELU_SLOPE) if route is None else route.act("%s%s.%d" % (tag, part, i), x, LRELU_SLOPE)
            elif op == "bn":
                if train:
                    B[key + "num_batches_tracked"] += 1
                x = F.batch_norm(x, B[key + "running_mean"], B[key + "running_var"], P[key + "weight"],
                                 P[key + "bias"], training=train, momentum=BN_MOMENTUM, eps=BN_EPS)
            elif op == "avgpool":
                x = F.avg_pool2d(x, 2)
            elif op == "upsample":
                if kw["mode"] == "nearest":
                    x = F.interpolate(x, scale_factor=2, mode="nearest")
                else:   # networks/ae_standard.py:68
                    x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
            elif op == "sigmoid":
                x = torch.sigmoid(x)
        return x

    def encode(self, x, train=True, route=None, tag=""):
        return self._run("enc", self.enc_prog, x, train, route, tag)

    def decode(self, z, train=True, route=None, tag=""):
        return self._run("dec", self.dec_prog, z, train, route, tag)

    def forward(self, x, train=True):
        return self.decode(self.encode(x, train), train)


def acdc_args(width=128, latent_width=32, depth=32, latent=128, colors=1):
    """Architecture dict of BASELINE configs C2/C3 (train_cardiac_aesr.py:216-218 + net_config.py:19-33)."""
    return dict(width=width, latent_width=latent_width, depth=depth, latent=latent, colors=colors,
                use_batchnorm=True, use_sigmoid=True, n_res_block=None, device="cpu")


def ae_standard_blocks(params, x):
    """networks/ae_standard.py:34-80 restated: BasicEncoderBlock(use_batchnorm=False, downsample=True) then BasicDecoderBlock
    (do_upsample=True).  ``params``: {"enc.conv2d_1.weight", ..., "dec.conv2d_2.bias"}.  Returns (mid, out)."""
    h = F.leaky_relu(F.conv2d(x, params["enc.conv2d_1.weight"], params["enc.conv2d_1.bias"], padding=1), 0.01)
    h = F.leaky_relu(F.conv2d(h, params["enc.conv2d_2.weight"], params["enc.conv2d_2.bias"], padding=1), 0.01)
    mid = F.avg_pool2d(h, 2)
    h = F.leaky_relu(F.conv2d(mid, params["dec.conv2d_1.weight"], params["dec.conv2d_1.bias"], padding=1), 0.01)
    h = F.leaky_relu(F.conv2d(h, params["dec.conv2d_2.weight"], params["dec.conv2d_2.bias"], padding=1), 0.01)
    out = F.interpolate(h, scale_factor=2, mode="bilinear", align_corners=False)
    return mid, out
