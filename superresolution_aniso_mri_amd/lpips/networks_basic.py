"""LPIPS ``PNetLin`` (VGG16 taps + unit-normalise + squared difference + 1x1 lin + spatial mean) on the HIP kernels.

Same constructor / forward contract and state_dict keys (``lin<k>.model.1.weight``, ``scaling_layer.shift|scale``,
``net.slice<k>.<idx>.weight|bias``) as the reference's lpips/networks_basic.py:19-110 and
lpips/pretrained_networks.py:97-135, restricted to what the ae_combined path uses: ``pnet_type='vgg'``, ``lpips=True``,
``spatial=False``, version '0.1', eval mode (dropout inactive, backbone frozen).

Both branches run as ONE batch of 2N images through the VGG stack (conv3x3+ReLU on the MFMA implicit-GEMM kernel,
conv1_1 on the small-Cin kernel with ScalingLayer -- and optionally the 2x-1 of perceptual.py -- folded into its
loader); each tap is one fused kernel; the backward pass touches only the branch that needs a gradient and computes
data gradients only (the backbone and the lin layers are constants of the loss)."""
import os

import numpy as np
import torch
import torch.nn as nn

from .. import _hip, engine
from .._hip import check, lib, ptr, stream
from .synthetic_vgg import VGG16_CFG, conv_indices, synthetic_vgg16_state

SHIFT = (-.030, -.088, -.188)
SCALE = (.458, .448, .450)
TAP_AFTER_CONV = (2, 4, 7, 10, 13)
_TRACE = None       # diagnostics (scripts/diag_vggfile_grad.py): a list collects (conv number, gradient of its pre-activation) per backward
_SLICE_BOUNDS = (4, 9, 16, 23, 30)      # features[x] belongs to slice k if x < bound[k] (pretrained_networks.py:107-116)


def spatial_average(in_tens, keepdim=True):
    return in_tens.mean([2, 3], keepdim=keepdim)


class ScalingLayer(nn.Module):
    def __init__(self):
        super(ScalingLayer, self).__init__()
        self.register_buffer("shift", torch.Tensor(SHIFT)[None, :, None, None])
        self.register_buffer("scale", torch.Tensor(SCALE)[None, :, None, None])

    def forward(self, inp):
        return (inp - self.shift) / self.scale


class NetLinLayer(nn.Module):
    """parameter holder of one 1x1 lin layer (``model.1.weight`` [1,C,1,1]); executed inside the fused tap kernel"""

    def __init__(self, chn_in, chn_out=1, use_dropout=False):
        super(NetLinLayer, self).__init__()
        layers = [nn.Dropout()] if use_dropout else []
        layers += [nn.Conv2d(chn_in, chn_out, 1, stride=1, padding=0, bias=False)]
        self.model = nn.Sequential(*layers)


class vgg16(nn.Module):
    """parameter holder with the reference's slice1..slice5 naming; requires_grad=False (frozen backbone)"""

    def __init__(self, requires_grad=False, pretrained=True, state_dict=None):
        super(vgg16, self).__init__()
        if requires_grad:
            raise NotImplementedError("pnet_tune=True (training the VGG backbone) is outside the ae_combined path")
        layers, cin = [], 3
        for v in VGG16_CFG:
            if v == "M":
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [nn.Conv2d(cin, v, 3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        self.N_slices = 5
        for k in range(5):
            setattr(self, "slice%d" % (k + 1), nn.Sequential())
        for x, m in enumerate(layers[:30]):
            k = next(i for i, b in enumerate(_SLICE_BOUNDS) if x < b)
            getattr(self, "slice%d" % (k + 1)).add_module(str(x), m)
        if state_dict is not None:
            self.load_features_state(state_dict)
        for p in self.parameters():
            p.requires_grad = False

    def load_features_state(self, sd):
        own = dict(self.named_parameters())
        for x in conv_indices():
            k = next(i for i, b in enumerate(_SLICE_BOUNDS) if x < b)
            for n in ("weight", "bias"):
                own["slice%d.%d.%s" % (k + 1, x, n)].data.copy_(sd["features.%d.%s" % (x, n)])

    def convs(self):
        out = []
        for k in range(5):
            for m in getattr(self, "slice%d" % (k + 1)):
                if isinstance(m, nn.Conv2d):
                    out.append(m)
        return out


class _LpipsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, mul, add, x0, x1):
        d, saved = net._forward_hip(x0, x1, mul, add, save=ctx.needs_input_grad[3])
        ctx.net, ctx.saved, ctx.mul = net, saved, mul
        return d

    @staticmethod
    def backward(ctx, gd):
        g0 = ctx.net._backward_hip(gd.contiguous().float().reshape(-1), ctx.saved, ctx.mul)
        ctx.saved = None
        return None, None, None, g0, None


# ScalingLayer + channel broadcast + conv1_1 folded into one single-channel convolution (PNetLin._conv1_1_folded); AESR_LPIPS_FOLD=0 keeps
# the 4-channel expansion + implicit GEMM
FOLD_CONV1_1 = os.environ.get("AESR_LPIPS_FOLD", "1") != "0"


class PNetLin(nn.Module):
    def __init__(self, pnet_type="vgg", pnet_rand=False, pnet_tune=False, use_dropout=True, spatial=False, version="0.1",
                 lpips=True, vgg_state_dict=None):
        super(PNetLin, self).__init__()
        if pnet_type not in ("vgg", "vgg16") or spatial or not lpips or version != "0.1":
            raise NotImplementedError("this build covers LPIPS v0.1 net-lin VGG, non-spatial (the ae_combined configuration)")
        self.pnet_type, self.pnet_tune, self.pnet_rand = pnet_type, pnet_tune, pnet_rand
        self.spatial, self.lpips, self.version = spatial, lpips, version
        self.scaling_layer = ScalingLayer()
        self.chns = [64, 128, 256, 512, 512]
        self.L_weights = [1., 1., 1., 1., 1.]
        self.L = len(self.chns)
        self.net = vgg16(pretrained=not pnet_rand, requires_grad=pnet_tune, state_dict=vgg_state_dict)
        for k, c in enumerate(self.chns):
            setattr(self, "lin%d" % k, NetLinLayer(c, use_dropout=use_dropout))
        self.lins = [getattr(self, "lin%d" % k) for k in range(self.L)]
        self._packed = None

    # ---- HIP execution ---------------------------------------------------------------------------------------
    def _ensure_packed(self):
        convs = self.net.convs()
        key = tuple((c.weight.data_ptr(), c.weight._version) for c in convs) + tuple(
            (l.model[-1].weight.data_ptr(), l.model[-1].weight._version) for l in self.lins)
        if self._packed is not None and self._packed["key"] == key:
            return self._packed
        pk = {"key": key, "fwd": [], "bwd": [], "fwd_wino": [], "bwd_wino": [], "lin": []}
        # conv1_1: 3 input channels padded to 4 (zero filter plane) so the MFMA kernel's 16-byte channel pieces apply
        w0 = convs[0].weight
        _hip.require_gpu_tensor(w0, "vgg weight")
        w4 = torch.zeros((w0.shape[0], 4, 3, 3), device=w0.device, dtype=torch.float32)
        w4[:, :3] = w0.detach()
        pk["w4"] = w4
        for c, w in [(None, w4)] + [(c, c.weight) for c in convs[1:]]:
            cin = w.shape[1]
            cout = w.shape[0]
            _hip.require_gpu_tensor(w, "vgg weight")
            # per direction: the Winograd-transformed filter where conv_wino_f32 applies (every layer but conv1_1), else the
            # implicit-GEMM packing
            for transpose, key in ((0, "fwd"), (1, "bwd")):
                wino = engine.wino_ok(cin, cout, 3, 1, transpose)
                if wino:
                    buf = torch.empty(lib.aesr_conv2d_wino_packed_floats(cout, cin, transpose), device=w.device)
                    job = (_hip.PackJob * 1)(_hip.PackJob(w.data_ptr(), buf.data_ptr(), cout, cin, 3, transpose))
                    check(lib.aesr_conv2d_wino_pack_many(job, 1, stream()), "aesr_conv2d_wino_pack_many")
                else:
                    buf = torch.empty(lib.aesr_conv2d_packed_floats(cout, cin, 3, transpose), device=w.device)
                    check(lib.aesr_conv2d_pack(ptr(w), ptr(buf), cout, cin, 3, transpose, stream()), "aesr_conv2d_pack")
                pk[key].append(buf)
                pk[key + "_wino"].append(wino)
        for k, l in enumerate(self.lins):
            if self.L_weights[k] != 1.0:
                raise NotImplementedError("L_weights != 1")
            pk["lin"].append(l.model[-1].weight.detach().reshape(-1).contiguous().float())
        self._packed = pk
        return pk

    def _conv1_1_folded(self, pk, mul, add):
        """ScalingLayer + the 1 -> 3 channel broadcast + conv1_1 (lpips/networks_basic.py:99-100, lpips/pretrained_networks.py:107 of the
        reference) as ONE single-channel 3x3 convolution: the three input planes are affine copies a_c x + b_c of the same slice, so
        conv1_1 of them is a 1 -> 64 convolution with weff[t][co] = sum_c W[co,c,t] a_c plus a per-tap bias beff[t][co] = sum_c W[co,c,t] b_c
        that counts only for taps inside the image (the zero padding applies to the 3-channel tensor) -- the stem-fold form of
        csrc/conv_thin.hip, bandwidth-bound.  Returns (folded [2 x 9 x 64] for aesr_stemconv_fwd, flipped filter [1,64,3,3] for the data
        gradient as a 64 -> 1 convolution); cached per (mul, add)."""
        cache = pk.setdefault("c11", {})
        if add is None:                       # backward: the filter only depends on ``mul``; any entry of the forward pass serves
            for (m, _), v in cache.items():
                if m == float(mul):
                    return v
            add = 0.0
        key = (float(mul), float(add))
        if key in cache:
            return cache[key]
        ca, cb = self._affine(mul, add)
        w0 = self.net.convs()[0].weight.detach().double()                     # [64, 3, 3, 3]
        a = torch.tensor(ca, dtype=torch.float64, device=w0.device)
        b = torch.tensor(cb, dtype=torch.float64, device=w0.device)
        weff = torch.einsum("ocyx,c->yxo", w0, a).reshape(9, -1)               # [t = 3 ky + kx][co]
        beff = torch.einsum("ocyx,c->yxo", w0, b).reshape(9, -1)
        folded = torch.cat([weff.reshape(-1), beff.reshape(-1)]).float().contiguous()
        wflip = weff.flip(0).t().reshape(1, -1, 3, 3).float().contiguous()      # [0, co, ky, kx] = weff[(2 - ky, 2 - kx)][co]
        if len(cache) > 8:
            cache.clear()
        cache[key] = (folded, wflip)
        return folded, wflip

    def _affine(self, mul, add):
        hc = self.__dict__.get("_scaling_host")
        if hc is None:          # one device read, cached: no host sync per call (and legal under HIP-graph capture)
            hc = self._scaling_host = (self.scaling_layer.shift.reshape(-1).tolist(), self.scaling_layer.scale.reshape(-1).tolist())
        sh, sc = hc
        return [mul / s for s in sc], [(add - h) / s for h, s in zip(sh, sc)]

    def _forward_hip(self, x0, x1, mul, add, save):
        """x0, x1: [B,H,W,1] fp32 on the GPU.  Returns (d [B,1,1,1], saved)."""
        pk = self._ensure_packed()
        convs = self.net.convs()
        B, H, W, _ = x0.shape
        x = torch.cat([x0, x1], dim=0)
        N = 2 * B
        acts, pool_in, taps = [], {}, []
        c0 = convs[0]
        cur = torch.empty((N, H, W, 64), device=x.device)
        if FOLD_CONV1_1:
            folded, _ = self._conv1_1_folded(pk, mul, add)
            engine._pb("thin_expand", 0.0, 4.0 * (x.numel() + cur.numel()))
            check(lib.aesr_stemconv_fwd(ptr(x), ptr(folded), ptr(c0.bias), ptr(cur), N, H, W, 64, 0, _hip.ACT_RELU, 0.0, stream()),
                  "aesr_stemconv_fwd(vgg conv1_1)")
            engine._pe()
        else:
            ca, cb = self._affine(mul, add)
            x4 = torch.empty((N, H, W, 4), device=x.device)
            check(lib.aesr_scale_expand_fwd(ptr(x), ptr(x4), N * H * W, _hip.float_array(ca), _hip.float_array(cb), stream()),
                  "aesr_scale_expand_fwd")
            check(lib.aesr_conv2d_fwd(ptr(x4), ptr(pk["fwd"][0]), ptr(c0.bias), ptr(cur), N, H, W, 4, 64, 3, 1, _hip.ACT_RELU, 0.0,
                                      stream()), "aesr_conv2d_fwd(vgg conv1_1)")
        acts.append(cur)
        nconv, h, w, cin = 1, H, W, 64
        partials, hws = [], []
        for v in VGG16_CFG[1:]:
            if v == "M":
                if h < 2 or w < 2:
                    raise ValueError("LPIPS-VGG needs images of at least 16x16 pixels")
                out = torch.empty((N, h // 2, w // 2, cin), device=x.device)
                engine._pb("maxpool2", 0.0, 4.0 * (cur.numel() + out.numel()))
                check(lib.aesr_maxpool2_fwd(ptr(cur), ptr(out), N, h, w, cin, stream()), "aesr_maxpool2_fwd")
                engine._pe()
                pool_in[nconv] = cur
                cur, h, w = out, h // 2, w // 2
                continue
            c = convs[nconv]
            out = torch.empty((N, h, w, v), device=x.device)
            if pk["fwd_wino"][nconv]:
                engine._pb(("wino", N, h, w, cin, v, 0), 2.0 * N * h * w * v * 9 * cin)
                ws, nws = engine.wino_workspace(cur, N, h, w, cin, v, 0)
                check(lib.aesr_conv2d_wino_fwd_ws(ptr(cur), ptr(pk["fwd"][nconv]), ptr(c.bias), ptr(out), ptr(ws), nws, N, h, w, cin, v,
                                                  _hip.ACT_RELU, 0.0, stream()), "aesr_conv2d_wino_fwd_ws(vgg)")
                engine._pe()
            else:
                nws = lib.aesr_conv2d_workspace_floats(N, h, w, cin, v, 3, 1)      # > 0: the small deep layers (conv4/5) get K-split
                ws = torch.empty((nws,), device=x.device) if nws else None
                check(lib.aesr_conv2d_fwd_ws(ptr(cur), ptr(pk["fwd"][nconv]), ptr(c.bias), ptr(out), ptr(ws), N, h, w, cin, v, 3, 1,
                                             _hip.ACT_RELU, 0.0, stream()), "aesr_conv2d_fwd_ws(vgg)")
            cur, cin = out, v
            nconv += 1
            acts.append(cur)
            if nconv in TAP_AFTER_CONV:
                k = len(taps)
                part = torch.empty((B, _hip.LPIPS_NCH), device=x.device)
                engine._pb("lpips_tap", 0.0, 4.0 * cur.numel())
                check(lib.aesr_lpips_tap_fwd(ptr(cur), ptr(pk["lin"][k]), ptr(part), B, h * w, v, stream()), "aesr_lpips_tap_fwd")
                engine._pe()
                taps.append((cur, h, w, v))
                partials.append(part)
                hws.append(h * w)
        d = torch.empty((B,), device=x.device)
        import ctypes
        parr = (ctypes.c_void_p * len(partials))(*[p.data_ptr() for p in partials])
        check(lib.aesr_lpips_finalize(parr, _hip.int_array(hws), len(partials), ptr(d), B, stream()), "aesr_lpips_finalize")
        saved = (acts, taps, (B, H, W)) if save else None
        if _TRACE is not None:
            _TRACE.append(("acts", list(acts)))
        return d.reshape(B, 1, 1, 1), saved

    def _backward_hip(self, gd, saved, mul):
        """gd [B]: dL/dd.  Returns the gradient w.r.t. x0 [B,H,W,1]."""
        pk = self._ensure_packed()
        convs = self.net.convs()
        acts, taps, (B, H, W) = saved
        dev = gd.device
        tap_of = {n: k for k, n in enumerate(TAP_AFTER_CONV)}       # conv count -> tap index
        # spatial size / channels of every conv output
        g = None                 # gradient w.r.t. the PRE-activation of conv `n` (1-based count), branch 0 only
        for n in range(13, 0, -1):
            a = acts[n - 1]
            _, h, w, c = a.shape
            if n in tap_of:
                k = tap_of[n]
                gtap = torch.empty((B, h, w, c), device=dev)
                engine._pb("lpips_tap", 0.0, 4.0 * (a.numel() + gtap.numel()))
                check(lib.aesr_lpips_tap_bwd(ptr(a), ptr(pk["lin"][k]), ptr(gd), ptr(gtap), B, h * w, c, stream()), "aesr_lpips_tap_bwd")
                engine._pe()
                if n == 13:
                    g = torch.empty_like(gtap)
                    engine._pb("act_bwd", 0.0, 12.0 * gtap.numel())
                    check(lib.aesr_act_bwd(ptr(gtap), ptr(a), ptr(g), gtap.numel(), _hip.ACT_RELU, 0.0, stream()), "aesr_act_bwd")
                    engine._pe()
                else:
                    # g currently holds the gradient w.r.t. the pooled tensor feeding conv n+1
                    dpre = torch.empty((B, h, w, c), device=dev)
                    engine._pb("maxpool2", 0.0, 4.0 * (g.numel() + 3 * dpre.numel()))
                    check(lib.aesr_maxpool2_bwd(ptr(g), ptr(a), ptr(gtap), ptr(dpre), B, h, w, c, 1, stream()), "aesr_maxpool2_bwd")
                    engine._pe()
                    g = dpre
            # now g = d/d(pre-activation of conv n); push it through conv n to its input
            if _TRACE is not None:
                _TRACE.append((n, g.clone()))
            if n == 1 and FOLD_CONV1_1:
                # data gradient of the folded layer with respect to the 1-channel slice: a 64 -> 1 convolution with the flipped filter
                _, wflip = self._conv1_1_folded(pk, mul, None)
                dx = torch.empty((B, H, W, 1), device=dev)
                engine._pb("thin_collapse", 0.0, 4.0 * (g.numel() + dx.numel()))
                check(lib.aesr_conv2d_cout1_fwd(ptr(g), ptr(wflip), None, ptr(dx), B, H, W, 64, _hip.ACT_NONE, 0.0, stream()),
                      "aesr_conv2d_cout1_fwd(vgg conv1_1 dgrad)")
                engine._pe()
                return dx
            if n == 1:
                ca, _ = self._affine(mul, 0.0)
                d4 = torch.empty((B, H, W, 4), device=dev)
                check(lib.aesr_conv2d_dgrad(ptr(g), ptr(pk["bwd"][0]), None, ptr(d4), B, H, W, 4, 64, 3, 1, 0, 0.0, stream()),
                      "aesr_conv2d_dgrad(vgg conv1_1)")
                dx = torch.empty((B, H, W, 1), device=dev)
                check(lib.aesr_scale_expand_bwd(ptr(d4), ptr(dx), B * H * W, _hip.float_array(ca), stream()), "aesr_scale_expand_bwd")
                return dx
            cv = convs[n - 1]
            prev = acts[n - 2]
            producer_is_pool = (n - 1) in tap_of           # conv n reads pool(tap layer n-1)
            mask = None if producer_is_pool else prev
            dxs = torch.empty((B, h, w, cv.in_channels), device=dev)
            if pk["bwd_wino"][n - 1]:
                engine._pb(("wino", B, h, w, cv.in_channels, cv.out_channels, 1), 2.0 * B * h * w * cv.in_channels * 9 * cv.out_channels)
                wsd, nwsd = engine.wino_workspace(g, B, h, w, cv.in_channels, cv.out_channels, 1)
                check(lib.aesr_conv2d_wino_dgrad_ws(ptr(g), ptr(pk["bwd"][n - 1]), ptr(mask), ptr(dxs), ptr(wsd), nwsd, B, h, w,
                                                    cv.in_channels, cv.out_channels, _hip.ACT_RELU if mask is not None else 0, 0.0, stream()),
                      "aesr_conv2d_wino_dgrad_ws(vgg)")
                engine._pe()
            else:
                nws = lib.aesr_conv2d_dgrad_workspace_floats(B, h, w, cv.in_channels, cv.out_channels, 3, 1)
                ws = torch.empty((nws,), device=dev) if nws else None
                check(lib.aesr_conv2d_dgrad_ws(ptr(g), ptr(pk["bwd"][n - 1]), ptr(mask), ptr(dxs), ptr(ws), B, h, w, cv.in_channels,
                                               cv.out_channels, 3, 1, _hip.ACT_RELU if mask is not None else 0, 0.0, stream()),
                      "aesr_conv2d_dgrad_ws(vgg)")
            g = dxs
        raise AssertionError("unreachable")

    def forward(self, in0, in1, retPerLayer=False, _affine=(1.0, 0.0)):
        """in0, in1: [N,1,H,W] or [N,3,H,W] in [-1,1] (or raw [0,1] images with ``_affine=(2,-1)``).  Returns [N,1,1,1]."""
        if retPerLayer:
            raise NotImplementedError("retPerLayer")
        if in0.shape != in1.shape or in0.dim() != 4:
            raise ValueError("LPIPS inputs must be two NCHW tensors of the same shape")
        if in0.shape[1] != 1:
            raise NotImplementedError("the HIP LPIPS path takes 1-channel slices (broadcast in ScalingLayer, SURVEY Q8)")
        swap = in1.requires_grad and not in0.requires_grad and torch.is_grad_enabled()
        if in0.requires_grad and in1.requires_grad and torch.is_grad_enabled():
            raise NotImplementedError("gradients w.r.t. both LPIPS inputs")
        a, b = (in1, in0) if swap else (in0, in1)      # (f0-f1)^2 is symmetric: the differentiated branch goes first
        N, _, H, W = a.shape

        def prep(t):
            t = t.float().reshape(N, H, W, 1) if t.is_contiguous() else t.float().contiguous().reshape(N, H, W, 1)
            return _hip.require_gpu_tensor(t, "LPIPS input")

        return _LpipsFn.apply(self, float(_affine[0]), float(_affine[1]), prep(a), prep(b))


def load_lin_weights(net, path=None):
    """LPIPS v0.1 linear calibration for VGG (data shipped as weights/v0.1/vgg_lin.npz)."""
    path = path or os.path.join(os.path.dirname(os.path.abspath(__file__)), "weights", "v0.1", "vgg_lin.npz")
    w = np.load(path)
    for k, l in enumerate(net.lins):
        l.model[-1].weight.data.copy_(torch.from_numpy(w["lin%d" % k]).reshape(1, -1, 1, 1))
