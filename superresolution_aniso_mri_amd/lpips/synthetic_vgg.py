"""Deterministic synthetic VGG16 backbone weights.  The reference downloads ImageNet weights through torchvision at
run time (lpips/pretrained_networks.py:100); there is no network here and the weights are not in the repository, so
unless the user supplies a local ``vgg16`` state_dict the backbone is filled from an integer hash (He-uniform scale).
LPIPS values computed with it are NOT the published LPIPS metric -- only its arithmetic."""
from collections import OrderedDict

import numpy as np
import torch

VGG16_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512]


def conv_indices():
    idx, out = 0, []
    for v in VGG16_CFG:
        if v == "M":
            idx += 1
        else:
            out.append(idx)
            idx += 2
    return out


def _hash_uniform(n, salt):
    i = np.arange(n, dtype=np.uint64)
    x = i ^ ((np.uint64(salt) * np.uint64(0x9E3779B1)) & np.uint64(0xFFFFFFFF))
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    return (x.astype(np.float64) / 2147483648.0 - 1.0).astype(np.float32)


def synthetic_vgg16_state():
    sd, cin = OrderedDict(), 3
    couts = [v for v in VGG16_CFG if v != "M"]
    for n, (idx, cout) in enumerate(zip(conv_indices(), couts)):
        bound = np.float32(np.sqrt(6.0 / (cin * 9)))
        sd["features.%d.weight" % idx] = torch.from_numpy((_hash_uniform(cout * cin * 9, 2 * n + 1) * bound).reshape(cout, cin, 3, 3).copy())
        sd["features.%d.bias" % idx] = torch.from_numpy((_hash_uniform(cout, 2 * n + 2) * np.float32(0.05)).copy())
        cin = cout
    return sd
