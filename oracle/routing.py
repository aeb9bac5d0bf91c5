"""Oracle (TEST INFRASTRUCTURE, see oracle/__init__.py): the NON-SMOOTH decisions of a pass -- the sign of every LeakyReLU / ReLU
input and the winner of every max-pool window -- recorded and, on request, FORCED to somebody else's.

Why: the ae_combined step is piecewise smooth.  Two correct evaluations that round differently agree to ~1e-7 in every forward
tensor, but an activation input within that rounding of zero (or two max-pool candidates within it of each other) may take the
other branch, and the derivative of that one element then differs by O(1): first-step gradients move by 1e-4 ... 2e-3 without any
kernel being wrong (profiles/r03_gradient_flip_analysis.txt, r04_percept_sensitivity.txt).  With this module a test can

  1. evaluate the oracle in fp64 with ITS OWN decisions and list where the HIP path decided otherwise, with the fp64 margin of each
     such decision (|pre-activation|, or winner minus runner-up) -- a margin of a few fp32 roundings proves a tie;
  2. evaluate the oracle in fp64 with the HIP path's decisions at exactly those places and compare gradients at a bound that a second
     flip or a real regression cannot hide behind.

The functions compute the SAME values as F.leaky_relu / F.relu / F.max_pool2d wherever no decision is forced (and a forced decision
changes a value only by the margin of the tie).  Follows networks/acai_vanilla.py:17,55-59 (LeakyReLU(0.01) behind the convolutions)
and lpips/pretrained_networks.py:107-116 (torchvision VGG16 ``features``: ReLU behind every convolution, MaxPool2d(2))."""
from collections import OrderedDict

import torch


class Routing(object):
    """``forced``: {name: bool mask (activations: input counts as positive) | int64 winner index 0..3 in scan order 00,01,10,11
    (max-pool windows)} in the oracle's NCHW layout.  ``seen[name]`` = (kind, fp values the decision was taken on, own decision)."""

    def __init__(self, forced=None):
        self.forced = forced or {}
        self.seen = OrderedDict()

    def act(self, name, x, slope):
        own = x > 0
        self.seen[name] = ("act", x.detach(), own)
        m = self.forced.get(name)
        m = own if m is None else m.to(torch.bool)
        return torch.where(m, x, slope * x)

    @staticmethod
    def windows(x):
        n, c, h, w = x.shape
        ho, wo = h // 2, w // 2
        return x[:, :, :2 * ho, :2 * wo].reshape(n, c, ho, 2, wo, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, c, ho, wo, 4)

    def maxpool(self, name, x):
        win = self.windows(x)
        own = win.argmax(-1)                       # first maximum in scan order: ATen's (and the HIP kernel's) tie rule
        self.seen[name] = ("pool", win.detach(), own)
        idx = self.forced.get(name)
        idx = own if idx is None else idx.to(torch.int64)
        return win.gather(-1, idx[..., None])[..., 0]


def differing_decisions(route, theirs, scale_floor=1e-30):
    """Decisions of ``theirs`` ({name: mask | winner}) that are not the oracle's own (``route.seen``), each with its margin in the oracle's
    arithmetic: activations |x| at the element; windows own maximum minus the value at their winner (0 = an exact tie, e.g. all zeros
    behind a ReLU: not a difference).  Returns a list of dicts sorted by relative margin, largest first: name, kind, index, margin,
    scale (rms of the layer's values), rel = margin / scale."""
    out = []
    for name, t in theirs.items():
        if name not in route.seen:
            raise KeyError("no decision named %s was taken by the oracle" % name)
        kind, vals, own = route.seen[name]
        scale = float(vals.double().pow(2).mean().sqrt()) + scale_floor
        if kind == "act":
            diff = (own != t.to(torch.bool)).nonzero()
            for ix in diff.tolist():
                m = abs(float(vals[tuple(ix)]))
                out.append(dict(name=name, kind=kind, index=tuple(ix), margin=m, scale=scale, rel=m / scale))
        else:
            ti = t.to(torch.int64)
            diff = (own != ti).nonzero()
            for ix in diff.tolist():
                w = vals[tuple(ix)]
                m = float(w.max() - w[int(ti[tuple(ix)])])
                if m == 0.0:
                    continue
                out.append(dict(name=name, kind=kind, index=tuple(ix), margin=m, scale=scale, rel=m / scale))
    out.sort(key=lambda d: -d["rel"])
    return out


def forced_only_where_different(route, theirs):
    """``theirs`` as a ``forced`` table for a second evaluation.  (Forcing every decision is the same thing as forcing the differing
    ones -- the others are the oracle's own; this helper just keeps the tables small.)"""
    return {name: t for name, t in theirs.items() if name in route.seen}
