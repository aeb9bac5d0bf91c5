"""Oracle (TEST INFRASTRUCTURE, see oracle/__init__.py): PyTorch-CPU fp32 restatement of the
LPIPS-VGG perceptual distance the reference uses for the synthesis loss.

Follows
  * lpips/perceptual.py:19-33        PerceptualLoss.forward: optional 2x-1, then model(target, pred)
  * lpips/networks_basic.py:63-91    PNetLin.forward; :93-100 ScalingLayer; :103-110 NetLinLayer
  * lpips/common.py:12-14            normalize_tensor (eps OUTSIDE the sqrt)
  * lpips/pretrained_networks.py:97-135  vgg16 wrapper, taps after features[3], [8], [15], [22], [29]

Third-party arithmetic absent from /root/reference: torchvision.models.vgg16 (unpinned; the
reference downloads ImageNet weights at run time, lpips/pretrained_networks.py:100).  Its
published architecture (VGG-16 configuration "D": 13 conv3x3+ReLU, 5 maxpool2) is restated in
`VGG16_CFG`; the backbone *weights* are unavailable offline, so parity for the backbone is
pinned only with the deterministic synthetic weights of `hash_vgg16_state()` -- parity
unpinned for real ImageNet weights.
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

# torchvision vgg16 `features` layout: ints = conv3x3(out channels)+ReLU, 'M' = MaxPool2d(2)
VGG16_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512]
TAP_AFTER_CONV = (2, 4, 7, 10, 13)          # 1-based conv counts at relu1_2 ... relu5_3
LPIPS_CHNS = (64, 128, 256, 512, 512)       # lpips/networks_basic.py:35
SHIFT = (-.030, -.088, -.188)               # lpips/networks_basic.py:96
SCALE = (.458, .448, .450)                  # lpips/networks_basic.py:97


def vgg16_feature_indices():
    """Index of every conv in torchvision's ``vgg16().features`` Sequential (conv, relu, [pool])."""
    idx, out = 0, []
    for v in VGG16_CFG:
        if v == "M":
            idx += 1
        else:
            out.append(idx)
            idx += 2
    return out                                # [0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28]


def _lowbias32(x):
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    return x


def hash_uniform(n, salt):
    """n floats in [-1, 1): u[i] = lowbias32(i XOR salt*0x9E3779B1) / 2^31 - 1 (float64 -> float32)."""
    i = np.arange(n, dtype=np.uint64)
    s = (np.uint64(salt) * np.uint64(0x9E3779B1)) & np.uint64(0xFFFFFFFF)
    h = _lowbias32(i ^ s)
    return (h.astype(np.float64) / 2147483648.0 - 1.0).astype(np.float32)


def hash_vgg16_state():
    """Deterministic synthetic VGG16 backbone (He-uniform scale so activations stay O(1) through
    13 layers).  Keys follow torchvision: ``features.<idx>.weight|bias``."""
    sd, cin = OrderedDict(), 3
    for n, (idx, cout) in enumerate(zip(vgg16_feature_indices(), [v for v in VGG16_CFG if v != "M"])):
        bound = np.sqrt(6.0 / (cin * 9))
        w = hash_uniform(cout * cin * 9, salt=2 * n + 1) * np.float32(bound)
        b = hash_uniform(cout, salt=2 * n + 2) * np.float32(0.05)
        sd["features.%d.weight" % idx] = torch.from_numpy(w.reshape(cout, cin, 3, 3).copy())
        sd["features.%d.bias" % idx] = torch.from_numpy(b.copy())
        cin = cout
    return sd


def vgg16_taps(x, vgg_sd, route=None, tag=""):
    """x [N,3,H,W] -> the five tap activations (after relu1_2, relu2_2, relu3_3, relu4_3, relu5_3).  ``route`` (oracle/routing.py,
    optional) records / forces the ReLU signs (``<tag>relu<n>``, n = 1-based conv count) and max-pool winners (``<tag>pool<n>``)."""
    taps, nconv, conv_idx = [], 0, vgg16_feature_indices()
    for v in VGG16_CFG:
        if v == "M":
            x = F.max_pool2d(x, 2) if route is None else route.maxpool("%spool%d" % (tag, nconv), x)
        else:
            i = conv_idx[nconv]
            x = F.conv2d(x, vgg_sd["features.%d.weight" % i], vgg_sd["features.%d.bias" % i], padding=1)
            nconv += 1
            x = F.relu(x) if route is None else route.act("%srelu%d" % (tag, nconv), x, 0.0)
            if nconv in TAP_AFTER_CONV:
                taps.append(x)
    return taps


def normalize_tensor(f, eps=1e-10):
    # lpips/common.py:12-14
    return f / (torch.sqrt(torch.sum(f ** 2, dim=1, keepdim=True)) + eps)


def lpips_head(taps0, taps1, lin_w, per_layer=False):
    """lpips/networks_basic.py:69-86 with lpips=True, spatial=False, dropout inactive (eval), L_weights=1.
    lin_w[k]: [1,C_k,1,1] (no bias).  Returns [N,1,1,1]."""
    res = []
    for f0, f1, w in zip(taps0, taps1, lin_w):
        d = (normalize_tensor(f0) - normalize_tensor(f1)) ** 2
        res.append(F.conv2d(d, w).mean([2, 3], keepdim=True))
    val = res[0]
    for r in res[1:]:
        val = val + r
    return (val, res) if per_layer else val


def scaling_layer(x):
    # lpips/networks_basic.py:99-100; a 1-channel image broadcasts to 3 channels here
    shift = torch.tensor(SHIFT, dtype=x.dtype)[None, :, None, None]
    scale = torch.tensor(SCALE, dtype=x.dtype)[None, :, None, None]
    return (x - shift) / scale


def lpips_distance(in0, in1, vgg_sd, lin_w, route=None, tag=""):
    """PNetLin.forward (version '0.1'): inputs already in [-1, 1].  ``route``: the decisions of the two branches go by
    ``<tag>in0/...`` and ``<tag>in1/...`` (perceptual_loss: in0 = target, in1 = pred)."""
    return lpips_head(vgg16_taps(scaling_layer(in0), vgg_sd, route, tag + "in0/"), vgg16_taps(scaling_layer(in1), vgg_sd, route, tag + "in1/"), lin_w)


def perceptual_loss(pred, target, vgg_sd, lin_w, normalize=True, route=None, tag=""):
    """lpips/perceptual.py:19-33 (note the argument swap: model.forward(target, pred))."""
    if normalize:
        target = 2 * target - 1
        pred = 2 * pred - 1
    return lpips_distance(target, pred, vgg_sd, lin_w, route, tag)
