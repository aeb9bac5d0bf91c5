"""-m gpu: the Laplacian-pyramid L1 loss (csrc/lap.hip through kwatsch/lap_pyramid_loss.py) against vectors of the reference's
LapLoss (tests/golden/laploss.npz), against the oracle on non-square / larger inputs, the transposed-operator identities the
backward relies on, and the trainer option ``use_laploss`` (reference kwatsch/base_trainer.py:47-56,183-196)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def hip():
    from superresolution_aniso_mri_amd import _hip
    assert torch.cuda.is_available()
    return _hip


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_laploss_vs_reference_vectors(tag):
    from kwatsch.lap_pyramid_loss import LapLoss, laplacian_pyramid
    rec = dict(np.load(os.path.join(GOLDEN, "laploss.npz")))
    x = torch.from_numpy(rec[tag + "/x"]).cuda().requires_grad_(True)
    t = torch.from_numpy(rec[tag + "/t"]).cuda()
    N, C, H, W = x.shape
    for k, p in enumerate(laplacian_pyramid(x, 3)):
        want = rec["%s/pyr%d" % (tag, k)]
        assert tuple(p.shape) == (N * C, H >> k, W >> k)
        np.testing.assert_allclose(p.detach().cpu().numpy().reshape(want.shape), want, rtol=0, atol=2e-6)
    loss = LapLoss(max_levels=3, channels=C, device="cuda")(x, t)
    loss.backward()
    assert abs(float(loss) - float(rec[tag + "/loss"])) < 1e-6
    # sign(a-b)/n terms: a pyramid difference within rounding of zero may take the other sign (none in these vectors)
    np.testing.assert_allclose(x.grad.cpu().numpy(), rec[tag + "/dx"], rtol=0, atol=1e-8 + 2e-6 * np.abs(rec[tag + "/dx"]).max())


@pytest.mark.parametrize("shape", [(2, 1, 160, 160), (3, 1, 24, 40), (1, 1, 8, 12)])
def test_laploss_vs_oracle_nonsquare_and_full_size(shape):
    """The reference's upsample() only runs on square inputs; the kernels and the oracle follow its arithmetic for any even size."""
    from kwatsch.lap_pyramid_loss import LapLoss
    from oracle import lap_oracle
    g = torch.Generator().manual_seed(sum(shape))
    xc = torch.rand(shape, generator=g).requires_grad_(True)
    tc = torch.rand(shape, generator=g)
    levels = 3 if min(shape[2:]) >= 24 else 2
    ref = lap_oracle.lap_loss(xc, tc, levels)
    ref.backward()
    x = xc.detach().cuda().requires_grad_(True)
    loss = LapLoss(max_levels=levels, channels=1)(x, tc.cuda())
    loss.backward()
    assert abs(float(loss) - float(ref)) < 2e-6 * abs(float(ref))
    assert rel_l2(x.grad, xc.grad) < 1e-5
    with pytest.raises(ValueError):
        LapLoss(max_levels=3, channels=1)(torch.rand(1, 1, 12, 12).cuda(), torch.rand(1, 1, 12, 12).cuda())     # 12 -> 6 -> 3 (odd)


def test_blur_adjoint_and_down_up_are_transposes(hip):
    """<G x, y> == <x, G^T y> for the reflect-padded filter (incl. the two border rows / columns), <down x, y> == <x, zero_insert y>."""
    L = hip.lib
    P, H, W = 2, 9, 14
    g = torch.Generator().manual_seed(5)
    x, y = torch.rand(P, H, W, generator=g).cuda(), torch.rand(P, H, W, generator=g).cuda()
    gx, gty = torch.empty_like(x), torch.empty_like(x)
    hip.check(L.aesr_lap_blur5(hip.ptr(x), None, hip.ptr(gx), P, H, W, 1.0, 0, hip.stream()), "blur")
    hip.check(L.aesr_lap_blur5(hip.ptr(y), None, hip.ptr(gty), P, H, W, 1.0, 1, hip.stream()), "blur^T")
    assert abs(float((gx.double() * y.double()).sum() - (x.double() * gty.double()).sum())) < 1e-4
    ref = torch.nn.functional.conv2d(torch.nn.functional.pad(x.cpu()[:, None], (2, 2, 2, 2), mode="reflect"),
                                     (torch.outer(torch.tensor([1., 4, 6, 4, 1]), torch.tensor([1., 4, 6, 4, 1])) / 256)[None, None])[:, 0]
    assert rel_l2(gx, ref) < 1e-6
    h, w = (H + 1) // 2, (W + 1) // 2
    d, z = torch.empty(P, h, w, device="cuda"), torch.empty(P, H, W, device="cuda")
    yy = torch.rand(P, h, w, generator=g).cuda()
    hip.check(L.aesr_lap_down2(hip.ptr(x), hip.ptr(d), P, H, W, hip.stream()), "down")
    hip.check(L.aesr_lap_zero_insert2(hip.ptr(yy), hip.ptr(z), P, h, w, H, W, hip.stream()), "zero_insert")
    assert torch.equal(d.cpu(), x.cpu()[:, ::2, ::2])
    assert abs(float((d.double() * yy.double()).sum() - (x.double() * z.double()).sum())) < 1e-5
    assert hip.lib.aesr_lap_blur5(hip.ptr(x), None, hip.ptr(gx), P, 2, W, 1.0, 0, hip.stream()) != 0       # H < 3: refused


@pytest.mark.parametrize("loss", ["mse", "perceptual"])
def test_trainer_with_laploss_vs_oracle(loss):
    """``use_laploss``: reconstruction loss = MSE + LapLoss (base_trainer.py:183-198); the synthesis loss gains the LapLoss term in
    its MSE form only (cardiac/trainer_ae.py:114-125).  One training step of the HIP trainer against the oracle arithmetic."""
    import torch.nn.functional as F
    from oracle import ae_oracle, lap_oracle, lpips_oracle
    from superresolution_aniso_mri_amd.data_synth import synthetic_batch
    from superresolution_aniso_mri_amd.kwatsch.get_trainer import get_trainer_dynamic
    from superresolution_aniso_mri_amd.networks.net_config import NetworkConfig
    cfg = dict(width=32, latent_width=8, depth=8, latent=16, colors=1, use_batchnorm=True, use_sigmoid=True)
    args = dict(model="ae_combined", dataset="ACDC", device="cuda", lr=1e-4, weight_decay=0.0, epochs=10, ex_loss_weight1=0.05,
                use_percept_loss=False, get_masks=False, use_loss_annealing=False, use_extra_latent_loss=False, epoch_threshold=100,
                ae_class="VanillaACAI", image_mix_loss_func=loss, vgg_weights="synthetic-hash", use_laploss=True, **cfg)
    for k, v in NetworkConfig("ae_combined", dataset="ACDC").architecture.items():
        args.setdefault(k, v)
    torch.manual_seed(11)
    tr = get_trainer_dynamic(args)
    assert tr.laploss is not None
    o = ae_oracle.OracleAE(cfg, init=False).load_state_dict({k: v.detach().cpu() for k, v in tr.model.state_dict().items()})
    batch = synthetic_batch(3, 32, 32, seed=21)
    x, btw = batch["image"], batch["slice_between"]
    tr.train(batch)
    z = o.encode(x, train=True)
    out = o.decode(z, train=True)
    s_mix = o.decode(0.5 * z[:3] + 0.5 * z[3:], train=True)
    dist, lap = F.mse_loss(out, x), lap_oracle.lap_loss(out, x)
    if loss == "mse":
        extra = lap_oracle.lap_loss(s_mix, btw) + F.mse_loss(btw, s_mix)
    else:
        lin = np.load(os.path.join(os.path.dirname(__file__), "..", "superresolution_aniso_mri_amd", "lpips", "weights", "v0.1", "vgg_lin.npz"))
        extra = lpips_oracle.perceptual_loss(btw, s_mix, lpips_oracle.hash_vgg16_state(),
                                             [torch.from_numpy(lin["lin%d" % k]).reshape(1, -1, 1, 1) for k in range(5)], normalize=True).mean()
    total = dist + lap + 0.05 * extra
    o.zero_grad()
    total.backward()
    assert abs(tr.losses["loss_laploss"][-1] - float(lap)) < 2e-5 * float(lap)
    assert abs(tr.losses["loss_ae_dist"][-1] - float(dist)) < 2e-5 * float(dist)
    assert abs(tr.losses["loss_ae"][-1] - float(total)) < 2e-5 * float(total)
    for k, p in tr.model.named_parameters():
        assert rel_l2(p.grad, o.params[k].grad) < 5e-4, k


def test_laploss_step_replays_from_graph():
    """The LapLoss kernels are capture-safe: the step with ``use_laploss`` replayed from a HIP graph equals the host-launched step."""
    from superresolution_aniso_mri_amd.data_synth import synthetic_batch
    from superresolution_aniso_mri_amd.kwatsch.get_trainer import get_trainer_dynamic
    from superresolution_aniso_mri_amd.networks.net_config import NetworkConfig
    args = dict(model="ae_combined", dataset="ACDC", device="cuda", lr=1e-4, weight_decay=0.0, epochs=10, ex_loss_weight1=0.05,
                use_percept_loss=False, get_masks=False, use_loss_annealing=False, use_extra_latent_loss=False, epoch_threshold=100,
                ae_class="VanillaACAI", image_mix_loss_func="mse", use_laploss=True, width=32, latent_width=8, depth=8, latent=16)
    for k, v in NetworkConfig("ae_combined", dataset="ACDC").architecture.items():
        args.setdefault(k, v)
    torch.manual_seed(5)
    eager = get_trainer_dynamic(dict(args))
    graphed = get_trainer_dynamic(dict(args))
    graphed.model.load_state_dict(eager.model.state_dict())
    graphed.enable_step_graph(eager_steps=1)
    for step in range(5):
        batch = synthetic_batch(3, 32, 32, seed=40 + step)
        eager.train(batch, keep_predictions=False)
        graphed.train(batch, keep_predictions=False)
    assert len(graphed._graphs) == 1
    for key in ("loss_ae", "loss_laploss", "loss_ae_dist_extra"):
        assert graphed.losses[key].floats() == eager.losses[key].floats(), key
    for (k, a), (_, b) in zip(eager.model.state_dict().items(), graphed.model.state_dict().items()):
        assert torch.equal(a, b), k
