"""-m gpu: the rest of the trainer surface callers touch: validate(), epoch hooks, best-model saving,
reconstruction-LPIPS mode (--use_percept_loss), plain `ae` trainer, interpolate helpers, and the BASELINE config 4 / 5
shapes (220x220 B=16, 256x256 B=8) through one training step against the CPU oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _args(tmp, **kw):
    from superresolution_aniso_mri_amd.networks.net_config import NetworkConfig
    a = dict(model="ae_combined", dataset="ACDC", device="cuda", lr=1e-4, weight_decay=0.0, epochs=5, width=32, latent_width=8, depth=8,
             latent=16, ex_loss_weight1=0.05, use_percept_loss=False, get_masks=False, use_loss_annealing=False,
             use_extra_latent_loss=False, epoch_threshold=0, ae_class="VanillaACAI", image_mix_loss_func="mse",
             vgg_weights="synthetic-hash", output_dir=str(tmp), dir_models=str(tmp), dir_images=str(tmp), log_tensorboard=False)
    a.update(kw)
    for k, v in NetworkConfig(a["model"], dataset=a["dataset"], ae_class=a["ae_class"]).architecture.items():
        a.setdefault(k, v)
    return a


def test_validate_and_epoch_hooks(tmp_path):
    from superresolution_aniso_mri_amd.data_synth import synthetic_batch
    from superresolution_aniso_mri_amd.kwatsch.get_trainer import get_trainer_dynamic
    torch.manual_seed(0)
    tr = get_trainer_dynamic(_args(tmp_path))
    tr.init_tensorboard(str(tmp_path))
    val = synthetic_batch(4, 32, 32, seed=99)
    for epoch in (1, 2, 3):
        tr.reset_losses()
        for it in range(2):
            tr.train(synthetic_batch(3, 32, 32, seed=epoch * 10 + it), keep_predictions=(it == 1))
        res = tr.validate(val)
        assert set(res) == {"img_grid_recons", "loss_ae"} and res["img_grid_recons"].shape[0] == 1
        assert len(tr.losses_test["loss_ae_dist_extra"]) == 1 and len(tr.losses_test["loss_latent_1"]) == 2
        tr.show_loss_on_tensorboard()
        tr.show_loss_on_tensorboard(eval_type="test")
        tr.generate_train_images(epoch=epoch, batch_item=synthetic_batch(3, 32, 32, seed=1))
        tr.end_epoch_processing(epoch=epoch, val_result_dict=res)
    assert tr.epoch == 3
    assert os.path.isfile(tmp_path / "2.models") and os.path.isfile(tmp_path / "losses_test.npz")
    assert len(tr.mean_losses["loss_ae"]) == 3 and tr.train_predictions["reconstruction"].shape == (6, 1, 32, 32)
    # eval wrappers return device tensors in NCHW logical layout
    z = tr.encode(val["image"])
    assert z.is_cuda and tuple(z.shape) == (8, 16, 8, 8) and tuple(tr.predict(val["image"]).shape) == (8, 1, 32, 32)


def test_validate_previews_whole_volumes_like_the_reference(tmp_path):
    """validate(validation_batch, image_dict=...) (kwatsch/base_trainer.py:67-99,149-162): the HIP trainer on two in-memory 4-D patients
    against what the reference's OWN trainer class returned on the CPU (tests/golden/val_volumes.npz): validation loss, the per-patient
    comparison grid (= the build's make_grid over the tensor the reference hands to torchvision's), alphas; end_epoch_processing writes
    one val_image_e###_p###.png per patient (:416-418)."""
    from superresolution_aniso_mri_amd.kwatsch.acai_utils import make_grid
    from superresolution_aniso_mri_amd.kwatsch.get_trainer import get_trainer_dynamic
    rec = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "val_volumes.npz")))
    tr = get_trainer_dynamic(_args(tmp_path, lr=1e-3, epoch_threshold=100))
    tr.model.load_state_dict({k[2:]: torch.from_numpy(v) for k, v in rec.items() if k.startswith("p/")})
    image_dict = {int(p): {"image": rec["p%d/image4d" % p], "patient_id": "patient%03d" % p, "spacing": np.array([8.0, 1.4, 1.4])}
                  for p in rec["patients"]}
    val = {"image": torch.from_numpy(rec["val/image"]), "slice_between": torch.from_numpy(rec["val/slice_between"])}
    res = tr.validate(val, image_dict=image_dict, frame_id=int(rec["frame_id"]), generate_images=True)
    assert set(res) == {"img_grid_recons", "loss_ae", "synthesized_vols", "alphas"}
    assert abs(float(res["loss_ae"]) - float(rec["val/loss_ae"])) <= 2e-5 * float(rec["val/loss_ae"])
    for p in image_dict:
        nrow, padding, pad_value = rec["p%d/grid_args" % p]
        want = make_grid(torch.from_numpy(rec["p%d/grid_input" % p]), int(nrow), padding=int(padding), pad_value=float(pad_value)).numpy()
        got = res["synthesized_vols"][p]
        assert got.shape == want.shape == (1, 7 * 34 + 2, int(nrow) * 34 + 2)
        assert np.abs(got - want).max() < 1e-5
        assert np.all(np.asarray(res["alphas"][p]) == 0.5) and tuple(res["alphas"][p].shape) == tuple(rec["p%d/alphas" % p].shape)
    # a frame beyond the last one means the last one (evaluate_image.py:50-51)
    vols, _ = tr._generate_val_volumes({3: image_dict[3]}, frame_id=8)
    assert vols[3].shape == res["synthesized_vols"][3].shape
    tr.end_epoch_processing(epoch=4, val_result_dict=res)
    try:
        import matplotlib  # noqa: F401
        assert os.path.isfile(tmp_path / "val_image_e004_p003.png") and os.path.isfile(tmp_path / "val_image_e004_p017.png")
        assert os.path.isfile(tmp_path / "val_recons_e004.png")
    except ImportError:
        pass


def test_plain_ae_and_reconstruction_lpips(tmp_path):
    from superresolution_aniso_mri_amd.data_synth import synthetic_batch
    from superresolution_aniso_mri_amd.kwatsch.get_trainer import get_trainer_dynamic
    torch.manual_seed(1)
    tr = get_trainer_dynamic(_args(tmp_path, model="ae", image_mix_loss_func=None))
    assert type(tr).__name__ == "AEBaseTrainer" and tr.percept_criterion is None
    b = synthetic_batch(2, 32, 32, seed=3)
    l0 = None
    for _ in range(5):
        tr.train(b)
        l0 = l0 or tr.losses["loss_ae"][-1]
    assert tr.losses["loss_ae"][-1] < l0 and tr.train_predictions["slice_inbetween_mix"].shape == (2, 1, 32, 32)
    with pytest.warns(UserWarning):
        tr2 = get_trainer_dynamic(_args(tmp_path, use_percept_loss=True, image_mix_loss_func="perceptual"))
    assert tr2.ae_loss_func == "perceptual"
    tr2.train(b)                                        # LPIPS on the reconstruction: gradient flows to the 2nd LPIPS argument
    assert np.isfinite(tr2.losses["loss_ae"][-1]) and all(p.grad is not None for p in tr2.model.parameters())


def test_interpolation_helpers(tmp_path):
    """kwatsch/acai_utils.py:41-103 of the reference (interpolate_2, create_interpol_grid): grid layout AND values -- the decoded latent
    mixes against the CPU oracle running the same recipe (encode in eval mode, z_a (1 - t) + z_b t for the interior points of
    linspace(0, 1, n + 2), decode in eval mode, torchvision's make_grid layout)."""
    from oracle import ae_oracle
    from superresolution_aniso_mri_amd.kwatsch.acai_utils import create_interpol_grid, interpolate_2, make_grid
    from superresolution_aniso_mri_amd.kwatsch.get_trainer import get_trainer_dynamic
    torch.manual_seed(3)
    tr = get_trainer_dynamic(_args(tmp_path), eval_mode=True)
    cfg = dict(width=32, latent_width=8, depth=8, latent=16, colors=1, use_batchnorm=True, use_sigmoid=True)
    oracle = ae_oracle.OracleAE(cfg, init=False).load_state_dict({k: v.detach().cpu() for k, v in tr.model.state_dict().items()})
    x = torch.rand(6, 1, 32, 32)

    def want(za, zb, first, last, n, nrow):
        with torch.no_grad():
            mixes = [oracle.decode(za * float(1 - t) + zb * float(t), train=False) for t in np.linspace(0., 1., n + 2)[1:-1]]
        return make_grid(torch.cat([first] + mixes + [last], dim=0), nrow, padding=2, pad_value=0.5).numpy()[0]

    with torch.no_grad():
        z = oracle.encode(x, train=False)
    g = interpolate_2(tr, x, num_interpol=3)
    assert g.ndim == 2 and g.shape[1] == 3 * 34 + 2 and g.shape[0] == 5 * 34 + 2        # 3 columns, 1 + 3 + 1 rows
    np.testing.assert_allclose(g, want(z[:3], z[-3:], x[:3], x[-3:], 3, 3), atol=2e-6)
    g2 = create_interpol_grid(tr, x[:, 0], num_interpol=2)
    assert g2.shape == (4 * 34 + 2, 5 * 34 + 2)
    np.testing.assert_allclose(g2, want(z[1:], z[:-1], x[1:], x[:-1], 2, 5), atol=2e-6)


@pytest.mark.parametrize("name,B,size,width,latent_width,dataset", [("c4_oasis", 16, 220, 64, 16, "OASIS"), ("c5_dhcp", 8, 256, 256, 64, "dHCP")])
def test_baseline_config_shapes_vs_oracle(tmp_path, name, B, size, width, latent_width, dataset):
    """BASELINE configs 4 / 5 (single rank): full-size step, forward quantities against the CPU oracle (MSE synthesis loss keeps
    the oracle fast; the LPIPS path is covered at small sizes)."""
    from oracle import ae_oracle, step_oracle
    from superresolution_aniso_mri_amd.data_synth import synthetic_batch
    from superresolution_aniso_mri_amd.kwatsch.get_trainer import get_trainer_dynamic
    cfg = dict(width=width, latent_width=latent_width, depth=32, latent=128, colors=1, use_batchnorm=True, use_sigmoid=True)
    torch.manual_seed(5)
    tr = get_trainer_dynamic(_args(tmp_path, dataset=dataset, ex_loss_weight1=0.001, lr=1e-5, **cfg))
    assert type(tr).__name__ == "AETrainerExtension1Brain"
    oracle = ae_oracle.OracleAE(cfg, init=False).load_state_dict({k: v.detach().cpu() for k, v in tr.model.state_dict().items()})
    ost = step_oracle.OracleStep(oracle, lr=1e-5, ex_loss_weight1=0.001, image_mix_loss_func="mse")
    batch = synthetic_batch(B, size, size, seed=11, brain=True)
    tr.train(batch)
    ref = ost.train(batch["image"], batch["slice_between"], batch["alpha_from"], batch["alpha_to"])
    for key, want in (("loss_ae", ref["loss_ae"]), ("loss_ae_dist_extra", ref["loss_ae_dist_extra"]), ("loss_latent_1", ref["loss_latent_1"])):
        assert abs(tr.losses[key][-1] - want) <= 5e-5 * abs(want), key
    out, s_mix = tr.train_predictions["reconstruction"], tr.train_predictions["slice_inbetween_mix"]
    assert tuple(out.shape) == (2 * B, 1, size, size)
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    assert rel(out, ref["out"]) < 1e-5 and rel(s_mix, ref["s_mix"]) < 1e-5
    d_ssim = abs(step_oracle.ssim(out[:4].numpy(), batch["image"][:4].numpy()) - step_oracle.ssim(ref["out"][:4].numpy(), batch["image"][:4].numpy()))
    assert d_ssim < 1e-3                                  # north_star: SSIM within 1e-3 of the reference path
    # parameters after the step: Adam(lr=1e-5) moved every weight by <= lr
    for (k, a), (_, b) in zip(tr.model.state_dict().items(), oracle.state_dict().items()):
        if a.dtype.is_floating_point:
            assert float((a.cpu() - b).abs().max()) <= 2.5e-5 + 1e-5 * float(b.abs().max()), k


def test_masked_mse_synthesis_loss(tmp_path):
    """--get_masks with the MSE synthesis loss (kwatsch/cardiac/trainer_ae.py:117-120): mean(mse_none * mask), in a full
    training step and as a value against the same expression on the CPU."""
    from superresolution_aniso_mri_amd.data_synth import synthetic_batch
    from superresolution_aniso_mri_amd.kwatsch.get_trainer import get_trainer_dynamic
    torch.manual_seed(0)
    tr = get_trainer_dynamic(_args(tmp_path, get_masks=True))
    mask = (torch.rand(3, 1, 32, 32, generator=torch.Generator().manual_seed(1)) > 0.5).float()
    ref = torch.rand(3, 1, 32, 32).cuda()
    syn = torch.rand(3, 1, 32, 32).cuda().requires_grad_(True)
    loss = tr.get_extra_image_loss(ref, syn, mask=mask)
    want = ((ref.cpu() - syn.detach().cpu()) ** 2 * mask).mean()
    assert abs(float(loss.detach()) - float(want)) < 1e-6 * float(want)
    loss.backward()
    assert float(syn.grad.abs().sum()) > 0
    batch = synthetic_batch(3, 32, 32, seed=4)
    batch["loss_mask"] = mask
    tr.train(batch)
    assert np.isfinite(tr.losses["loss_ae"][-1])


@pytest.mark.parametrize("name,B,size,width,latent_width,dataset", [("c4_oasis", 16, 220, 64, 16, "OASIS"), ("c5_dhcp", 8, 256, 256, 64, "dHCP")])
def test_baseline_config_shapes_with_lpips_vs_oracle(tmp_path, name, B, size, width, latent_width, dataset):
    """BASELINE configs[3] / configs[4] as BASELINE.json states them -- WITH the LPIPS-VGG synthesis loss (lambda 0.001, per-sample
    mixing coefficients of the brain trainer), full size, one rank: first step against the CPU oracle (losses, predictions, first-step
    gradients, parameters after Adam)."""
    from oracle import ae_oracle, lpips_oracle, step_oracle
    from superresolution_aniso_mri_amd.data_synth import synthetic_batch
    from superresolution_aniso_mri_amd.kwatsch.get_trainer import get_trainer_dynamic
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = dict(width=width, latent_width=latent_width, depth=32, latent=128, colors=1, use_batchnorm=True, use_sigmoid=True)
    torch.manual_seed(5)
    with pytest.warns(UserWarning):
        tr = get_trainer_dynamic(_args(tmp_path, dataset=dataset, ex_loss_weight1=0.001, lr=1e-5, image_mix_loss_func="perceptual", **cfg))
    assert type(tr).__name__ == "AETrainerExtension1Brain" and tr.percept_criterion is not None
    oracle = ae_oracle.OracleAE(cfg, init=False).load_state_dict({k: v.detach().cpu() for k, v in tr.model.state_dict().items()})
    lin = np.load(os.path.join(root, "superresolution_aniso_mri_amd", "lpips", "weights", "v0.1", "vgg_lin.npz"))
    ost = step_oracle.OracleStep(oracle, lr=1e-5, ex_loss_weight1=0.001, image_mix_loss_func="perceptual", vgg_sd=lpips_oracle.hash_vgg16_state(),
                                 lin_w=[torch.from_numpy(lin["lin%d" % k]).reshape(1, -1, 1, 1) for k in range(5)])
    batch = synthetic_batch(B, size, size, seed=11, brain=True)
    tr.train(batch)
    ref = ost.train(batch["image"], batch["slice_between"], batch["alpha_from"], batch["alpha_to"])
    for key, want in (("loss_ae", ref["loss_ae"]), ("loss_ae_dist_extra", ref["loss_ae_dist_extra"]), ("loss_latent_1", ref["loss_latent_1"])):
        assert abs(tr.losses[key][-1] - want) <= 5e-5 * abs(want), key
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    assert rel(tr.train_predictions["reconstruction"], ref["out"]) < 1e-5 and rel(tr.train_predictions["slice_inbetween_mix"], ref["s_mix"]) < 1e-5
    num = den = 0.0
    for k, p in tr.model.named_parameters():
        g, r = p.grad.detach().double().cpu(), oracle.params[k].grad.double()
        num += float((g - r).pow(2).sum())
        den += float(r.pow(2).sum())
    assert (num / den) ** 0.5 < 2.5e-4
    for (k, a), (_, b) in zip(tr.model.state_dict().items(), oracle.state_dict().items()):
        if a.dtype.is_floating_point:
            assert float((a.cpu() - b).abs().max()) <= 2.5e-5 + 1e-5 * float(b.abs().max()), k


def test_vgg_weights_file_is_loaded(tmp_path):
    """--vgg_weights FILE (lpips/dist_model.py:24-25 of this build; replaces torchvision.models.vgg16(pretrained=True) of the reference's
    lpips/pretrained_networks.py:100): a torchvision-keyed ``features.N.weight / .bias`` state_dict on disk is what the HIP LPIPS
    computes with -- distances and gradient equal the oracle's with the SAME weights, and differ from the synthetic backbone's."""
    from oracle import lpips_oracle
    from superresolution_aniso_mri_amd.lpips.perceptual import PerceptualLoss
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g = torch.Generator().manual_seed(21)
    sd = {}
    for idx, (cin, cout) in zip(lpips_oracle.vgg16_feature_indices(), [(3, 64), (64, 64), (64, 128), (128, 128), (128, 256), (256, 256), (256, 256),
                                                                      (256, 512), (512, 512), (512, 512), (512, 512), (512, 512), (512, 512)]):
        sd["features.%d.weight" % idx] = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
        sd["features.%d.bias" % idx] = torch.randn(cout, generator=g) * 0.05
    f = str(tmp_path / "vgg16_features.pth")
    torch.save(sd, f)
    crit = PerceptualLoss(model="net-lin", net="vgg", use_gpu=True, gpu_ids=[0], vgg_weights=f, device="cuda")
    with pytest.warns(UserWarning):
        synth = PerceptualLoss(model="net-lin", net="vgg", use_gpu=True, gpu_ids=[0], vgg_weights="synthetic-hash", device="cuda")
    a = torch.rand(3, 1, 48, 40, generator=g)
    b = torch.rand(3, 1, 48, 40, generator=g)
    bd = b.cuda().requires_grad_(True)
    from superresolution_aniso_mri_amd.lpips import networks_basic as nb
    nb._TRACE = []                      # the activations of the stack: its ReLU / max-pool decisions (oracle/routing.py)
    try:
        d = crit(a.cuda(), bd, normalize=True)
        d.mean().backward()
        torch.cuda.synchronize()
        acts = [t for t in nb._TRACE if t[0] == "acts"][0][1]
    finally:
        nb._TRACE = None
    lin = np.load(os.path.join(root, "superresolution_aniso_mri_amd", "lpips", "weights", "v0.1", "vgg_lin.npz"))
    lin_w = [torch.from_numpy(lin["lin%d" % k]).reshape(1, -1, 1, 1) for k in range(5)]
    br = b.clone().requires_grad_(True)
    ref = lpips_oracle.perceptual_loss(a, br, sd, lin_w, normalize=True)
    ref.mean().backward()
    np.testing.assert_allclose(d.detach().cpu().numpy().ravel(), ref.detach().numpy().ravel(), rtol=2e-5)
    # The gradient, judged against the oracle evaluated in fp64.  With these weights ONE max-pool window of relu2_2 (image 1, rows 20-21,
    # columns 8-9, channel 35) holds two candidates closer than fp32 rounding, and the folded conv1_1 rounds them the other way round than
    # fp64 does: the window's gradient goes to the other pixel, which moves a 14 x 16-pixel patch of the input gradient (140 of 5 760
    # elements, 3.0e-3 of the gradient's norm); every stage before that window agrees to 1e-5 and every element outside the patch to
    # 4e-6 (scripts/diag_vggfile_grad.py, profiles/r03_lpips_maxpool_flip.txt).  So: the bulk must agree to rounding, a flipped
    # window or two may show in the norm.
    b64 = b.double().clone().requires_grad_(True)
    lpips_oracle.perceptual_loss(a.double(), b64, {k: v.double() for k, v in sd.items()}, [w.double() for w in lin_w], normalize=True).mean().backward()
    err = (bd.grad.double().cpu() - b64.grad).abs()
    gmax = float(b64.grad.abs().max())
    assert float(err.flatten().quantile(0.9)) < 2e-5 * gmax
    assert float(err.norm() / b64.grad.norm()) < 1e-2
    # ... and since round 6 that account is CHECKED, not assumed: the decisions the HIP stack took on the differentiated branch, the ones fp64 takes
    # otherwise (each must be a tie), and the gradient against fp64 evaluated under the HIP stack's decisions
    from oracle import routing
    dec = {}
    for n, act in enumerate(acts, start=1):
        a0 = act[:3].detach().permute(0, 3, 1, 2).cpu()
        dec["in0/relu%d" % n] = a0 > 0
        if n in lpips_oracle.TAP_AFTER_CONV[:-1]:
            dec["in0/pool%d" % n] = routing.Routing.windows(a0).argmax(-1)
    sd64, lin64 = {k: v.double() for k, v in sd.items()}, [w.double() for w in lin_w]
    r_own = routing.Routing()
    lpips_oracle.perceptual_loss(a.double(), b.double(), sd64, lin64, normalize=True, route=r_own)
    diffs = routing.differing_decisions(r_own, dec)
    assert len(diffs) <= 6 and all(x["rel"] <= 2e-5 for x in diffs), diffs
    bf = b.double().clone().requires_grad_(True)
    lpips_oracle.perceptual_loss(a.double(), bf, sd64, lin64, normalize=True, route=routing.Routing(dec)).mean().backward()
    assert float((bd.grad.double().cpu() - bf.grad).norm() / bf.grad.norm()) < 3e-5, (len(diffs), float((bd.grad.double().cpu() - bf.grad).norm() / bf.grad.norm()))
    assert float((br.grad.double() - b64.grad).norm() / b64.grad.norm()) < 1e-4           # the fp32 CPU oracle itself: 1.8e-6
    d_syn = synth(a.cuda(), b.cuda(), normalize=True)
    assert float((d_syn.cpu() - d.detach().cpu()).abs().max()) > 1e-3 * float(d.detach().abs().max())


def test_model_sr_and_load_caisr_round_trip(tmp_path):
    """The second ("SR") model of the eval wrappers (kwatsch/base_trainer.py:213-214,298-300 here; reference :325-336,358-367) through the
    loader's own route (kwatsch/get_trainer.py:42-55: ``model_nbr_sr`` -> models/<nbr>.models -> ``model_sr`` + ``load_caisr``):
    ``use_sr_model=True`` routes encode / decode through it, and generate_hr_volumes synthesises with it."""
    from superresolution_aniso_mri_amd.data_synth import synthetic_batch
    from superresolution_aniso_mri_amd.generate_hr_volumes import create_super_volume
    from superresolution_aniso_mri_amd.kwatsch.common import saveExperimentSettings
    from superresolution_aniso_mri_amd.kwatsch.get_trainer import get_trainer_dynamic
    src = tmp_path / "exper"
    os.makedirs(src / "models")
    a = _args(src)
    saveExperimentSettings(a, str(src / "settings.yaml"))
    torch.manual_seed(0)
    main = get_trainer_dynamic(dict(a))
    main.train(synthetic_batch(3, 32, 32, seed=0))
    main.save_models(str(src / "models" / "3.models"), 3)
    torch.manual_seed(1)                                  # the SR checkpoint: a different, further trained model
    donor = get_trainer_dynamic(dict(a))
    for it in range(3):
        donor.train(synthetic_batch(3, 32, 32, seed=10 + it))
    donor.save_models(str(src / "models" / "5.models"), 5)
    tr, args2 = get_trainer_dynamic(src_path=str(src), model_nbr=3, model_nbr_sr=5, eval_mode=True)
    assert tr.model_sr is not None and tr.model_sr is not tr.model and args2["latent"] == a["latent"]
    for (k, v), (_, w) in zip(tr.model_sr.state_dict().items(), donor.model.state_dict().items()):
        assert torch.equal(v.cpu(), w.cpu()), k
    for (k, v), (_, w) in zip(tr.model.state_dict().items(), main.model.state_dict().items()):
        assert torch.equal(v.cpu(), w.cpu()), k
    x = torch.rand(4, 1, 32, 32)
    z_sr, z_main, z_donor = tr.encode(x, use_sr_model=True), tr.encode(x), donor.encode(x)
    assert torch.allclose(z_sr, z_donor, rtol=1e-6, atol=1e-7) and not torch.allclose(z_sr, z_main, rtol=1e-3, atol=1e-4)
    assert torch.allclose(tr.decode(z_sr, use_sr_model=True), donor.decode(z_donor), rtol=1e-6, atol=1e-7)
    hr_sr = create_super_volume(tr, x, [0.5], use_original=False)["upsampled_image"]          # synthesis goes through the SR model
    hr_donor = create_super_volume(donor, x, [0.5], use_original=False)["upsampled_image"]
    np.testing.assert_allclose(hr_sr.numpy(), hr_donor.numpy(), rtol=1e-6, atol=1e-7)
    # no SR checkpoint asked for: use_sr_model falls back to the main model (reference _use_sr_model)
    tr0, _ = get_trainer_dynamic(src_path=str(src), model_nbr=3, eval_mode=True)
    assert tr0.model_sr is None and torch.equal(tr0.encode(x, use_sr_model=True), tr0.encode(x))


def test_cosine_annealing_lr_steps_vs_oracle(tmp_path):
    """--use_lr_scheduler (kwatsch/base_trainer.py:58-62 here, reference :18-22: CosineAnnealingLR(opt, lr_iter_max, eta_min=0), stepped
    once per training step): three steps against the oracle with the same scheduler; the step graph stays off (per-step learning rates)."""
    from oracle import ae_oracle, step_oracle
    from superresolution_aniso_mri_amd.data_synth import synthetic_batch
    from superresolution_aniso_mri_amd.kwatsch.get_trainer import get_trainer_dynamic
    cfg = dict(width=32, latent_width=8, depth=8, latent=16, colors=1, use_batchnorm=True, use_sigmoid=True)
    torch.manual_seed(3)
    tr = get_trainer_dynamic(_args(tmp_path, lr=1e-3, use_lr_scheduler=True, lr_iter_max=4))
    tr.enable_step_graph(eager_steps=0)
    oracle = ae_oracle.OracleAE(cfg, init=False).load_state_dict({k: v.detach().cpu() for k, v in tr.model.state_dict().items()})
    ost = step_oracle.OracleStep(oracle, lr=1e-3, ex_loss_weight1=0.05, image_mix_loss_func="mse")
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(ost.opt, 4, eta_min=0, last_epoch=-1)
    lrs = []
    for it in range(3):
        b = synthetic_batch(3, 32, 32, seed=50 + it)
        tr.train(b, keep_predictions=False)
        ref = ost.train(b["image"], b["slice_between"])
        sched.step()
        lrs.append(tr.opt_ae.param_groups[0]["lr"])
        assert abs(tr.losses["loss_ae"][-1] - ref["loss_ae"]) <= 5e-5 * abs(ref["loss_ae"]), it
        assert abs(lrs[-1] - ost.opt.param_groups[0]["lr"]) < 1e-12
    assert lrs[0] > lrs[1] > lrs[2] > 0 and not getattr(tr, "_graphs", None)
    for (k, a), (_, b_) in zip(tr.model.state_dict().items(), oracle.state_dict().items()):
        if a.dtype.is_floating_point:
            assert float((a.cpu() - b_).abs().max()) <= 3 * 2 * 1e-3 + 1e-6, k
