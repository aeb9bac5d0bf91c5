"""Diagnostic for tests/test_gpu_trainer_api.py::test_vgg_weights_file_is_loaded: where does the HIP LPIPS input gradient differ from fp64?"""
import os, sys, tempfile
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import lpips_oracle
from superresolution_aniso_mri_amd.lpips.perceptual import PerceptualLoss

H, W = int(sys.argv[1]), int(sys.argv[2])
bias_scale = float(sys.argv[3]) if len(sys.argv) > 3 else 0.05
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = torch.Generator().manual_seed(21)
sd = {}
for idx, (cin, cout) in zip(lpips_oracle.vgg16_feature_indices(), [(3, 64), (64, 64), (64, 128), (128, 128), (128, 256), (256, 256), (256, 256),
                                                                  (256, 512), (512, 512), (512, 512), (512, 512), (512, 512), (512, 512)]):
    sd["features.%d.weight" % idx] = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    sd["features.%d.bias" % idx] = torch.randn(cout, generator=g) * bias_scale
f = os.path.join(tempfile.mkdtemp(), "vgg16_features.pth")
torch.save(sd, f)
crit = PerceptualLoss(model="net-lin", net="vgg", use_gpu=True, gpu_ids=[0], vgg_weights=f, device="cuda")
a = torch.rand(3, 1, H, W, generator=g)
b = torch.rand(3, 1, H, W, generator=g)
lin = np.load(os.path.join(root, "superresolution_aniso_mri_amd", "lpips", "weights", "v0.1", "vgg_lin.npz"))
lin_w = [torch.from_numpy(lin["lin%d" % k]).reshape(1, -1, 1, 1) for k in range(5)]
b64 = b.double().clone().requires_grad_(True)
lpips_oracle.perceptual_loss(a.double(), b64, {k: v.double() for k, v in sd.items()}, [w.double() for w in lin_w], normalize=True).mean().backward()
from superresolution_aniso_mri_amd.lpips import networks_basic as nb
import torch.nn.functional as F
sd64 = {k: v.double() for k, v in sd.items()}


def taps_with_pres(x, pres):
    taps, nconv, conv_idx = [], 0, lpips_oracle.vgg16_feature_indices()
    for v in lpips_oracle.VGG16_CFG:
        if v == "M":
            x = F.max_pool2d(x, 2)
        else:
            i = conv_idx[nconv]
            pre = F.conv2d(x, sd64["features.%d.weight" % i], sd64["features.%d.bias" % i], padding=1)
            if pres is not None:
                pre.retain_grad()
                pres.append(pre)
            x = F.relu(pre)
            nconv += 1
            if nconv in lpips_oracle.TAP_AFTER_CONV:
                taps.append(x)
    return taps


bb = b.double().clone().requires_grad_(True)
pres = []
t_b = taps_with_pres(lpips_oracle.scaling_layer(2 * bb - 1), pres)
t_a = taps_with_pres(lpips_oracle.scaling_layer(2 * a.double() - 1), None)
lpips_oracle.lpips_head(t_a, t_b, [w.double() for w in lin_w]).mean().backward()
print("fp64 reference with hooks against the oracle's gradient: %.2e" % float((bb.grad - b64.grad).abs().max()))
ref = {n + 1: p.grad.permute(0, 2, 3, 1) for n, p in enumerate(pres)}
ref[0] = bb.grad
nb._TRACE = []
bd = b.cuda().requires_grad_(True)
crit(a.cuda(), bd, normalize=True).mean().backward()
torch.cuda.synchronize()
for n, g in [t for t in nb._TRACE if t[0] != 'acts'] + [(0, bd.grad)]:
    r = ref[n]
    d = (g.double().cpu() - r).abs()
    bad = (d > 1e-3 * float(r.abs().max())).nonzero()
    print("conv %2d %s: rel l2 %.3e; elements off by > 1e-3 of the max: %d%s" % (n, tuple(g.shape), float(d.norm() / r.norm()), len(bad),
          "" if len(bad) == 0 else "; dim0 %s dim1 %s dim2 %s dim3 %s" % tuple(sorted(set(bad[:, k].tolist()))[:16] for k in range(4))))

# is the forward pass bit-reproducible under a different allocator state?
runs = []
for poison in (None, "1e3", "nan"):
    if poison:
        junk = [torch.full((n,), float(poison), device="cuda") for n in [1 << 28] + [1 << k for k in range(8, 20)] * 8]
        del junk
    nb._TRACE = []
    with torch.no_grad():
        crit(a.cuda(), b.cuda(), normalize=True)
    torch.cuda.synchronize()
    runs.append([t.clone() for t in nb._TRACE[0][1]])
x = taps_with_pres  # fp64 activations of both branches for the rounding distance
for k in range(13):
    d1 = (runs[0][k] - runs[1][k]).abs()
    d2 = (runs[0][k] - runs[2][k]).abs()
    print("act %2d %s: runs differ in %d / %d elements (max %.3e); nan-poison run: %d differ, %d nan" % (
        k + 1, tuple(runs[0][k].shape), int((d1 > 0).sum()), int((d2 > 0).sum()), float(d1.max()), int((d2 > 0).sum()), int(torch.isnan(runs[2][k]).sum())))
