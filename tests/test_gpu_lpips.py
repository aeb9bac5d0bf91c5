"""-m gpu: LPIPS-VGG on the HIP kernels against the vectors generated from the reference's lpips modules
(tests/golden/lpips_head.npz, lpips_full_*.npz; synthetic hash backbone + the reference's local lin weights).
Tolerances (fp32): distances rel 1e-5, gradients rel-L2 1e-4."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def test_maxpool_fwd_bwd():
    from superresolution_aniso_mri_amd import _hip as hip
    g = torch.Generator().manual_seed(1)
    N, C, H, W = 2, 64, 11, 14
    x = F.relu(torch.randn(N, C, H, W, generator=g)).requires_grad_(True)       # many exact ties at 0
    out = F.max_pool2d(x, 2)
    gout = torch.randn(out.shape, generator=g)
    gadd = torch.randn(x.shape, generator=g)
    out.backward(gout)
    xd, god, gad = nhwc(x.detach()).cuda(), nhwc(gout).cuda(), nhwc(gadd).cuda()
    od = torch.empty((N, H // 2, W // 2, C), device="cuda")
    hip.check(hip.lib.aesr_maxpool2_fwd(hip.ptr(xd), hip.ptr(od), N, H, W, C, hip.stream()), "fwd")
    assert torch.equal(od.cpu(), nhwc(out.detach()))
    dx = torch.empty((N, H, W, C), device="cuda")
    hip.check(hip.lib.aesr_maxpool2_bwd(hip.ptr(god), hip.ptr(xd), hip.ptr(gad), hip.ptr(dx), N, H, W, C, 1, hip.stream()), "bwd")
    ref = (x.grad + gadd) * (x.detach() > 0)
    assert rel_l2(dx.permute(0, 3, 1, 2), ref) < 1e-6


def test_tap_kernels_vs_reference_head():
    from superresolution_aniso_mri_amd import _hip as hip
    rec = dict(np.load(os.path.join(GOLDEN, "lpips_head.npz")))
    lin = np.load(os.path.join(os.path.dirname(GOLDEN), "..", "superresolution_aniso_mri_amd", "lpips", "weights", "v0.1", "vgg_lin.npz"))
    B = 2
    parts, hws, keep = [], [], []
    gd = torch.ones(B, device="cuda")
    for k in range(5):
        # the golden gradients are w.r.t. branch 1; (f0-f1)^2 is symmetric, so feed it as OUR branch 0
        fa, fb = torch.from_numpy(rec["f1_%d" % k]), torch.from_numpy(rec["f0_%d" % k])
        _, C, h, w = fa.shape
        f = torch.cat([nhwc(fa), nhwc(fb)]).cuda()
        lw = torch.from_numpy(lin["lin%d" % k]).cuda()
        part = torch.empty((B, hip.LPIPS_NCH), device="cuda")
        hip.check(hip.lib.aesr_lpips_tap_fwd(hip.ptr(f), hip.ptr(lw), hip.ptr(part), B, h * w, C, hip.stream()), "tap_fwd")
        res = part.sum(1).cpu() / (h * w)
        np.testing.assert_allclose(res.numpy(), rec["res_%d" % k].reshape(-1), rtol=1e-5)
        gf = torch.empty((B, h, w, C), device="cuda")
        hip.check(hip.lib.aesr_lpips_tap_bwd(hip.ptr(f), hip.ptr(lw), hip.ptr(gd), hip.ptr(gf), B, h * w, C, hip.stream()), "tap_bwd")
        assert rel_l2(gf.permute(0, 3, 1, 2), rec["g1_%d" % k]) < 1e-4, k
        parts.append(part), hws.append(h * w), keep.append((f, lw))
    import ctypes
    parr = (ctypes.c_void_p * 5)(*[p.data_ptr() for p in parts])
    d = torch.empty(B, device="cuda")
    hip.check(hip.lib.aesr_lpips_finalize(parr, hip.int_array(hws), 5, hip.ptr(d), B, hip.stream()), "finalize")
    np.testing.assert_allclose(d.cpu().numpy(), rec["val"].reshape(-1), rtol=1e-5)


@pytest.mark.parametrize("name", ["lpips_full_2x32x32.npz", "lpips_full_1x48x40.npz"])
def test_perceptual_loss_vs_reference(name):
    from superresolution_aniso_mri_amd.lpips.perceptual import PerceptualLoss
    rec = dict(np.load(os.path.join(GOLDEN, name)))
    with pytest.warns(UserWarning, match="SYNTHETIC backbone"):
        crit = PerceptualLoss(model="net-lin", net="vgg", use_gpu=True, gpu_ids=[0], vgg_weights="synthetic-hash")
    ref = torch.from_numpy(rec["ref"]).cuda()
    syn = torch.from_numpy(rec["syn"]).cuda().requires_grad_(True)
    d = crit(ref, syn, normalize=True)          # trainer call order: (reference, synthesized)
    assert tuple(d.shape) == (ref.shape[0], 1, 1, 1)
    d.mean().backward()
    np.testing.assert_allclose(d.detach().cpu().numpy(), rec["d"], rtol=2e-5)
    assert rel_l2(syn.grad, rec["dsyn"]) < 1e-4
    # argument order / which side carries the gradient must not matter: (a-b)^2 is symmetric
    syn2 = torch.from_numpy(rec["syn"]).cuda().requires_grad_(True)
    d2 = crit(syn2, ref, normalize=True)
    d2.mean().backward()
    np.testing.assert_allclose(d2.detach().cpu().numpy(), rec["d"], rtol=2e-5)
    assert rel_l2(syn2.grad, rec["dsyn"]) < 1e-4
    # and against the CPU oracle with pre-normalised inputs (normalize=False path)
    from oracle import lpips_oracle
    lin = np.load(os.path.join(os.path.dirname(GOLDEN), "..", "superresolution_aniso_mri_amd", "lpips", "weights", "v0.1", "vgg_lin.npz"))
    lw = [torch.from_numpy(lin["lin%d" % k]).reshape(1, -1, 1, 1) for k in range(5)]
    a, b = torch.from_numpy(rec["ref"]) * 2 - 1, torch.from_numpy(rec["syn"]) * 2 - 1
    want = lpips_oracle.perceptual_loss(a, b, lpips_oracle.hash_vgg16_state(), lw, normalize=False)
    with torch.no_grad():
        got = crit(a.cuda(), b.cuda(), normalize=False)
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=2e-5)
