"""not-gpu: the CPU oracle reproduces the vectors generated from the reference's own modules
(oracle/make_golden.py).  This is what pins the oracle; GPU parity tests then compare HIP vs oracle."""
import glob
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ae_oracle, lpips_oracle, step_oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SMALL = dict(width=32, latent_width=8, depth=8, latent=16, colors=1, use_batchnorm=True, use_sigmoid=True)


def _load(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def _sd(rec, prefix):
    return {k[len(prefix):]: torch.from_numpy(v) for k, v in rec.items() if k.startswith(prefix)}


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "ae_small_*.npz"))))
def test_ae_small_matches_reference(path):
    rec = dict(np.load(path))
    cname = os.path.basename(path).split("_")[2]
    ae = ae_oracle.OracleAE(SMALL, ae_class=cname, init=False).load_state_dict(_sd(rec, "p0/"))
    x = torch.from_numpy(rec["x"]).requires_grad_(True)
    z = ae.encode(x, train=True)
    out = ae.decode(z, train=True)
    loss = F.mse_loss(out, torch.from_numpy(rec["tgt"])) + 0.1 * (z ** 2).mean()
    loss.backward()
    # same torch build, same ops -> only thread-count summation noise is tolerated
    np.testing.assert_allclose(z.detach().numpy(), rec["z"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out.detach().numpy(), rec["out"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(x.grad.numpy(), rec["dx"], rtol=1e-4, atol=1e-7)
    for k, p in ae.params.items():
        np.testing.assert_allclose(p.grad.numpy(), rec["grad/" + k], rtol=2e-4, atol=1e-7, err_msg=k)
    for k, v in _sd(rec, "p1/").items():
        np.testing.assert_allclose(ae.buffers[k].numpy(), v.numpy(), rtol=1e-5, atol=1e-7, err_msg=k)
    with torch.no_grad():
        out_eval = ae.forward(x.detach(), train=False)
    np.testing.assert_allclose(out_eval.numpy(), rec["out_eval"], rtol=1e-5, atol=1e-6)


def test_init_is_rng_exact():
    rec = _load("ae_init_acdc.npz")
    torch.manual_seed(892372)
    ae = ae_oracle.OracleAE(ae_oracle.acdc_args())
    assert sum(p.numel() for p in ae.parameters()) == int(rec["nparams"]) == 443777
    for k, p in ae.params.items():
        assert np.array_equal(p.detach().flatten()[:4].numpy(), rec["head/" + k]), k
        assert abs(p.double().sum().item() - float(rec["sum/" + k])) <= 1e-9 * max(1.0, float(rec["abs/" + k])), k


def test_acdc_probe():
    rec = _load("ae_acdc_probe.npz")
    torch.manual_seed(892372)
    ae = ae_oracle.OracleAE(ae_oracle.acdc_args())
    image, _ = step_oracle.synthetic_triplets(1, 160, 160, seed=892372)
    z = ae.encode(image, train=True)
    out = ae.decode(z, train=True)
    assert z.shape == (2, 128, 40, 40) and out.shape == (2, 1, 160, 160)
    loss = F.mse_loss(out, image)
    loss.backward()
    assert abs(loss.item() - float(rec["loss"])) < 1e-6
    np.testing.assert_allclose(out.detach().flatten()[rec["out_idx"]].numpy(), rec["out_val"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(z.detach().flatten()[rec["z_idx"]].numpy(), rec["z_val"], rtol=1e-4, atol=1e-5)
    for k, p in ae.params.items():
        assert abs(p.grad.double().norm().item() - float(rec["gnorm/" + k])) <= 1e-4 * float(rec["gnorm/" + k]) + 1e-9, k


def _lin_w():
    path = os.path.join(os.path.dirname(GOLDEN), "..", "superresolution_aniso_mri_amd", "lpips", "weights", "v0.1",
                        "vgg_lin.npz")
    w = np.load(path)
    return [torch.from_numpy(w["lin%d" % k]).reshape(1, -1, 1, 1) for k in range(5)]


def test_lpips_head():
    rec = _load("lpips_head.npz")
    f0 = [torch.from_numpy(rec["f0_%d" % k]) for k in range(5)]
    f1 = [torch.from_numpy(rec["f1_%d" % k]).requires_grad_(True) for k in range(5)]
    val, res = lpips_oracle.lpips_head(f0, f1, _lin_w(), per_layer=True)
    val.sum().backward()
    np.testing.assert_allclose(val.detach().numpy(), rec["val"], rtol=1e-5)
    for k in range(5):
        np.testing.assert_allclose(res[k].detach().numpy(), rec["res_%d" % k], rtol=1e-5)
        np.testing.assert_allclose(f1[k].grad.numpy(), rec["g1_%d" % k], rtol=1e-4, atol=1e-9)


@pytest.mark.parametrize("name", ["lpips_full_2x32x32.npz", "lpips_full_1x48x40.npz"])
def test_lpips_full(name):
    rec = _load(name)
    vgg = lpips_oracle.hash_vgg16_state()
    ref = torch.from_numpy(rec["ref"])
    syn = torch.from_numpy(rec["syn"]).requires_grad_(True)
    d = lpips_oracle.perceptual_loss(ref, syn, vgg, _lin_w(), normalize=True)
    d.mean().backward()
    assert d.shape == (ref.shape[0], 1, 1, 1)
    np.testing.assert_allclose(d.detach().numpy(), rec["d"], rtol=1e-4)
    np.testing.assert_allclose(syn.grad.numpy(), rec["dsyn"], rtol=1e-3, atol=1e-8)
    taps = lpips_oracle.vgg16_taps(lpips_oracle.scaling_layer(2 * syn.detach() - 1), vgg)
    for k, t in enumerate(taps):
        assert abs(t.double().norm().item() - float(rec["tap%d_norm" % k])) < 1e-4 * float(rec["tap%d_norm" % k])


STEP_CASES = {   # fixture tag -> (oracle keyword overrides, lr, steps); every fixture is the reference's OWN trainer class run on the CPU
    "cardiac_lpips": (dict(image_mix_loss_func="perceptual"), 1e-3, 3),
    "brain_lpips": (dict(image_mix_loss_func="perceptual"), 1e-3, 3),
    "cardiac_mse": (dict(image_mix_loss_func="mse"), 1e-3, 3),
    "cardiac_percept": (dict(image_mix_loss_func="perceptual", recon_loss="perceptual"), 1e-3, 3),
    "ae_plain": (dict(image_mix_loss_func="mse", plain=True), 1e-3, 3),
    "cardiac_mse_s3": (dict(image_mix_loss_func="mse"), 1e-3, 3),
    "cardiac_mse_lr1e-5": (dict(image_mix_loss_func="mse"), 1e-5, 3),
}


def small_cfg(tag):
    return dict(SMALL, latent_width=4) if tag.endswith("_s3") else SMALL


@pytest.mark.parametrize("tag", sorted(STEP_CASES))
def test_train_steps(tag):
    kw, lr, nsteps = STEP_CASES[tag]
    rec = _load("step_k3_%s.npz" % tag)
    ae = ae_oracle.OracleAE(small_cfg(tag), init=False).load_state_dict(_sd(rec, "p0/"))
    st = step_oracle.OracleStep(ae, lr=lr, ex_loss_weight1=0.05, vgg_sd=lpips_oracle.hash_vgg16_state(), lin_w=_lin_w(), **kw)
    af = torch.from_numpy(rec["alpha_from"]) if "alpha_from" in rec else None
    at = torch.from_numpy(rec["alpha_to"]) if "alpha_to" in rec else None
    keys = [str(k) for k in rec["loss_keys"]]
    for step in range(nsteps):
        r = st.train(torch.from_numpy(rec["image_%d" % step]), torch.from_numpy(rec["between_%d" % step]), af, at)
        got = [r[k] for k in keys]
        # step 0 is a pure fwd comparison; later steps sit behind Adam updates of size ~lr*sign(g), where the
        # sign of a near-zero gradient is summation-order noise (thread count), so they get a looser bound at lr 1e-3
        np.testing.assert_allclose(got, rec["losses"][step], rtol=2e-5 if (step == 0 or lr < 1e-4) else 5e-3)
        if step == 0:
            np.testing.assert_allclose(r["z"].numpy(), rec["z_0"], rtol=1e-4, atol=1e-6)
            np.testing.assert_allclose(r["s_mix"].numpy(), rec["s_mix_0"], rtol=1e-4, atol=1e-6)
            np.testing.assert_allclose(r["out"].numpy(), rec["out_0"], rtol=1e-4, atol=1e-6)
            for k, p in ae.params.items():
                g, want = p.grad.double(), torch.from_numpy(rec["grad0/" + k]).double()
                assert float((g - want).norm()) <= 2e-4 * float(want.norm()) + 1e-12, k
    sd = ae.state_dict()
    for k, v in _sd(rec, "p%d/" % nsteps).items():
        # Adam moves every weight by ~lr*sign(g) per step; where g is summation-order noise around 0 the sign is
        # arbitrary, so a few elements may legitimately differ by up to 2*lr per step.  Bound both the bulk
        # (>=97% within 0.2 lr) and the worst case (3 steps * 2 * lr).
        a, b = sd[k].numpy().astype(np.float64), v.numpy().astype(np.float64)
        diff = np.abs(a - b)
        assert diff.max() <= nsteps * 2 * lr + 1e-6, k
        assert (diff > 0.2 * lr + 1e-3 * np.abs(b)).sum() <= max(1, 0.03 * diff.size), k


def test_loss_annealing_steps():
    """use_loss_annealing (kwatsch/cardiac/trainer_ae.py:80-83, kwatsch/base_trainer.py:456-459): the reference trainer ran one
    step in each of 4 epochs; its weight table and the logged lambda * loss sequence are reproduced."""
    rec = _load("step_k4_cardiac_anneal.npz")
    w = step_oracle.annealing_weights(4, 0.05)
    np.testing.assert_allclose(w, rec["loss_weights"], rtol=1e-12)
    assert w[0] > w[1] > w[2] > w[3] > 0
    ae = ae_oracle.OracleAE(SMALL, init=False).load_state_dict(_sd(rec, "p0/"))
    st = step_oracle.OracleStep(ae, lr=1e-3, ex_loss_weight1=0.05, image_mix_loss_func="mse")
    keys = [str(k) for k in rec["loss_keys"]]
    for step in range(4):
        r = st.train(torch.from_numpy(rec["image_%d" % step]), torch.from_numpy(rec["between_%d" % step]), lam=float(w[step]))
        np.testing.assert_allclose([r[k] for k in keys], rec["losses"][step], rtol=2e-5 if step == 0 else 5e-3)


@pytest.mark.parametrize("tag", ["c2", "c3"])
def test_step_probe_baseline_size(tag):
    """BASELINE configs[1] / [2] at their own size (12 triplets 160x160): one step of the reference's AETrainerEndToEnd, stored
    as a probe, reproduced by the oracle from the same seed."""
    from superresolution_aniso_mri_amd.data_synth import synthetic_batch
    rec = _load("step_probe_%s.npz" % tag)
    torch.manual_seed(892372)
    ae = ae_oracle.OracleAE(ae_oracle.acdc_args())
    for k, v in ae.state_dict().items():
        assert abs(float(v.double().sum()) - float(rec["init_sum/" + k])) <= 1e-9 * max(1.0, abs(float(rec["init_sum/" + k]))), k
    st = step_oracle.OracleStep(ae, lr=1e-5, ex_loss_weight1=0.05, image_mix_loss_func="mse" if tag == "c2" else "perceptual",
                                vgg_sd=lpips_oracle.hash_vgg16_state(), lin_w=_lin_w())
    batch = synthetic_batch(12, 160, 160, seed=892372)
    r = st.train(batch["image"], batch["slice_between"])
    got = [r[k] for k in ("loss_ae", "loss_ae_dist", "loss_ae_dist_extra", "loss_latent_1")]
    np.testing.assert_allclose(got, rec["losses"], rtol=2e-5)
    np.testing.assert_allclose(r["out"].flatten()[rec["out_idx"]].numpy(), rec["out_val"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(r["s_mix"].flatten()[rec["s_idx"]].numpy(), rec["s_val"], rtol=1e-4, atol=1e-6)
    for k, p in ae.params.items():
        assert abs(p.grad.double().norm().item() - float(rec["gnorm/" + k])) <= 2e-4 * float(rec["gnorm/" + k]) + 1e-12, k


def test_supervolume():
    rec = _load("supervolume.npz")
    ae = ae_oracle.OracleAE(SMALL, init=False).load_state_dict(_sd(rec, "p/"))
    hr = step_oracle.create_super_volume(ae, torch.from_numpy(rec["vol"]), rec["alpha_range"], use_original=True)
    assert hr.shape == rec["hr"].shape == ((5 - 1) * (3 + 1) + 1, 32, 32)
    np.testing.assert_allclose(hr.numpy(), rec["hr"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("tag", ["default", "inbetween_rem", "inbetween_even", "downsample_only"])
def test_supervolume_eval_protocol(tag):
    """oracle restatement of evaluate/common.py:134-235 (sub-sampling, remainder slices, alphas) against outputs of the
    reference's own function (tests/golden/supervolume_eval.npz, oracle/make_golden.py:gen_supervolume_eval)."""
    rec = _load("supervolume_eval.npz")
    ae = ae_oracle.OracleAE(SMALL, init=False).load_state_dict(_sd(rec, "p/"))
    ds = int(rec[tag + "/downsample_steps"])
    hr, alphas = step_oracle.create_super_volume_eval(
        ae, torch.from_numpy(rec[tag + "/vol"]), rec[tag + "/alpha_range"] if tag + "/alpha_range" in rec else None,
        use_original=True, downsample_steps=None if ds < 0 else ds,
        generate_inbetween_slices=bool(rec[tag + "/generate_inbetween_slices"]))
    assert hr.shape == rec[tag + "/hr"].shape
    np.testing.assert_allclose(hr.numpy(), rec[tag + "/hr"], rtol=1e-5, atol=1e-6)
    assert np.array_equal(alphas.numpy(), rec[tag + "/pred_alphas_first"])


def test_validation_volume_previews():
    """oracle restatement of ``_generate_val_volumes`` -> ``evaluate_image`` -> ``create_compare_image`` (kwatsch/base_trainer.py:149-162,
    evaluate/evaluate_image.py:37-106) against what the reference's OWN trainer returned from ``validate(image_dict=...)`` on two in-memory
    4-D patients (tests/golden/val_volumes.npz, oracle/make_golden.py:gen_val_volumes): the tensor it hands to make_grid and its nrow."""
    rec = _load("val_volumes.npz")
    ae = ae_oracle.OracleAE(SMALL, init=False).load_state_dict(_sd(rec, "p/"))
    for p in rec["patients"]:
        orig, synth, stack, k = step_oracle.val_volume_compare(ae, rec["p%d/image4d" % p], int(rec["frame_id"]), SMALL["width"])
        np.testing.assert_array_equal(orig, rec["p%d/orig" % p])
        np.testing.assert_allclose(synth, rec["p%d/synth" % p], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(stack, rec["p%d/grid_input" % p], rtol=1e-5, atol=2e-6)
        assert [k, 2, 0.5] == list(rec["p%d/grid_args" % p])
        assert np.all(rec["p%d/alphas" % p] == 0.5)


@pytest.mark.parametrize("tag", ["acai_combined", "acai"])
def test_acai_step_oracle_vs_reference_modules(tag):
    """oracle/step_oracle.OracleACAIStep (kwatsch/trainer_acai.py:46-127) against the step restated around the reference's own
    VanillaACAI / Discriminator modules (tests/golden/step_acai*.npz): losses, first-step gradients of both networks, parameters."""
    rec = _load("step_%s.npz" % tag)
    ae = ae_oracle.OracleAE(SMALL, init=False).load_state_dict(_sd(rec, "p0/"))
    critic = ae_oracle.OracleAE(SMALL, init=False).load_state_dict({k.replace("encoder.", "enc."): v for k, v in _sd(rec, "d0/").items()})
    st = step_oracle.OracleACAIStep(ae, critic, lr=1e-3, lamb_reg_acai=0.5, ex_loss_weight1=0.05, combined=(tag == "acai_combined"),
                                    image_mix_loss_func="mse")
    af, at = torch.from_numpy(rec["alpha_from"]), torch.from_numpy(rec["alpha_to"])
    for step in range(len(rec["losses"])):
        r = st.train(torch.from_numpy(rec["image_%d" % step]), torch.from_numpy(rec["between_%d" % step]), af, at,
                     torch.from_numpy(rec["alpha_%d" % step]))
        got = [r["loss_ae"], r["loss_disc"], r["loss_ae_dist"], r["loss_ae_dist_extra"], r["loss_latent_1"]]
        np.testing.assert_allclose(got, rec["losses"][step], rtol=1e-5 if step == 0 else 2e-3)
        if step == 0:
            for k, p in ae.params.items():
                np.testing.assert_allclose(p.grad.numpy(), rec["grad0/" + k], rtol=1e-4, atol=1e-7, err_msg=k)
            for k, p in critic.params.items():
                np.testing.assert_allclose(p.grad.numpy(), rec["dgrad0/" + k.replace("enc.", "encoder.")], rtol=1e-4, atol=1e-7, err_msg=k)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_laploss_oracle_vs_reference(tag):
    """oracle/lap_oracle.py against the reference's LapLoss (tests/golden/laploss.npz: pyramid levels, loss, input gradient)."""
    from oracle import lap_oracle
    rec = _load("laploss.npz")
    x, t = torch.from_numpy(rec[tag + "/x"]).requires_grad_(True), torch.from_numpy(rec[tag + "/t"])
    for k, p in enumerate(lap_oracle.laplacian_pyramid(x)):
        assert np.array_equal(p.detach().numpy(), rec["%s/pyr%d" % (tag, k)])
    loss = lap_oracle.lap_loss(x, t)
    loss.backward()
    assert float(loss.detach()) == float(rec[tag + "/loss"]) and np.array_equal(x.grad.numpy(), rec[tag + "/dx"])


@pytest.mark.parametrize("tag", ["a", "b"])
def test_ae_standard_blocks_oracle_vs_reference(tag):
    """The oracle's restatement of the ae_standard encoder/decoder blocks (AvgPool without BatchNorm, bilinear x2 upsample,
    networks/ae_standard.py:34-80) against vectors produced by the reference's own block modules."""
    from oracle import ae_oracle
    rec = dict(np.load(os.path.join(GOLDEN, "ae_standard_blocks_%s.npz" % tag)))
    params = {k[2:]: torch.from_numpy(v).requires_grad_(True) for k, v in rec.items() if k.startswith("p/")}
    x = torch.from_numpy(rec["x"]).requires_grad_(True)
    mid, out = ae_oracle.ae_standard_blocks(params, x)
    tgt = torch.from_numpy(rec["tgt"])
    loss = (out * tgt).mean() + 0.5 * (out ** 2).mean()
    loss.backward()
    assert np.allclose(mid.detach().numpy(), rec["mid"], rtol=1e-5, atol=1e-6)
    assert np.allclose(out.detach().numpy(), rec["out"], rtol=1e-5, atol=1e-6)
    assert abs(loss.item() - float(rec["loss"])) < 1e-6 * max(1.0, abs(float(rec["loss"])))
    assert np.allclose(x.grad.numpy(), rec["dx"], rtol=1e-4, atol=1e-7)
    for k, p in params.items():
        assert np.allclose(p.grad.numpy(), rec["grad/" + k], rtol=1e-4, atol=1e-6), k


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_augmentation_oracle_vs_reference_transforms(tag):
    """oracle/augment_oracle.py (pad, centre crop, random crop, sigmoid intensity, rot90; random numbers in the reference's order)
    reproduces the outputs of the reference's transform classes bit for bit (tests/golden/augment_acdc.npz)."""
    from oracle import augment_oracle as ao
    rec = dict(np.load(os.path.join(GOLDEN, "augment_acdc.npz")))
    aug, width, seed = [int(v) for v in rec[tag + "/cfg"]]
    rs = np.random.RandomState(seed)
    outs = [ao.augment_triplet(t, aug, width, rs)[0] for t in rec[tag + "/in"]]
    assert np.array_equal(np.stack(outs), rec[tag + "/out"])
    batch = ao.assemble_batch(outs)
    B = len(outs)
    assert batch["image"].shape == (2 * B, 1, width, width) and np.array_equal(batch["image"][B:, 0], rec[tag + "/out"][:, 1])


def test_vif_oracle_vs_reference_functions():
    """oracle/vif_oracle.py (numpy restatement incl. the Gaussian filter) against tests/golden/vif.npz = outputs of the reference's own
    ``evaluate.metrics.compute_vif_for_batch`` and ``evaluate.vifvec.vifp_mscale`` (uint8 / float32 / float64 slices).  Everything in
    front of the logarithms is the same sequence of IEEE operations: bit-equal is the expectation, 1e-12 the bound."""
    from oracle import vif_oracle as vo
    rec = dict(np.load(os.path.join(GOLDEN, "vif.npz")))
    tags = sorted({k.split("/")[0] for k in rec if k.endswith("/vif_u8")})
    assert len(tags) >= 5
    for t in tags:
        a, b = rec[t + "/ref"], rec[t + "/dist"]
        ds = int(rec[t + "/downsample_steps"])
        mean, per = vo.compute_vif_for_batch(a, b, downsample_steps=None if ds < 0 else ds)
        assert abs(mean - float(rec[t + "/vif_batch"])) < 1e-12, t
        u8 = np.array([vo.vifp_mscale(vo.to_uint8(a[k]), vo.to_uint8(b[k])) for k in range(a.shape[0])])
        assert np.abs(u8 - rec[t + "/vif_u8"]).max() < 1e-12, t
        f32 = np.array([vo.vifp_mscale(a[k], b[k]) for k in range(a.shape[0])])
        assert np.abs(f32 - rec[t + "/vif_f32"]).max() < 1e-12, t
        f64 = np.array([vo.vifp_mscale(a[k].astype(np.float64), b[k].astype(np.float64)) for k in range(a.shape[0])])
        assert np.abs(f64 - rec[t + "/vif_f64"]).max() < 1e-12, t
        # the uint8 arithmetic (what the evaluation runs) is NOT the float one: the fixture would catch a "cleaned up" port
        assert np.abs(rec[t + "/vif_u8"] - rec[t + "/vif_f32"]).max() > 1e-3, t
    assert abs(vo.compute_vif_for_batch(rec["img/ref"], rec["img/dist"])[0] - float(rec["img/vif_batch"])) < 1e-12
    assert abs(vo.compute_vif_for_batch(rec["img/ref"], rec["img/ref"])[0] - float(rec["same/vif_batch"])) < 1e-12
    assert np.isnan(rec["black/vif_batch"]) and np.isnan(vo.compute_vif_for_batch(np.zeros((2, 24, 24), np.float32), np.zeros((2, 24, 24), np.float32))[0])


# ---- the non-smooth decisions of the step (oracle/routing.py) ----------------------------------------------------------------------------
def _fixture_step(tag):
    kw, lr, _ = STEP_CASES[tag]
    rec = _load("step_k3_%s.npz" % tag)

    def make():
        ae = ae_oracle.OracleAE(small_cfg(tag), init=False).load_state_dict(_sd(rec, "p0/"))
        return step_oracle.OracleStep(ae, lr=lr, ex_loss_weight1=0.05, vgg_sd=lpips_oracle.hash_vgg16_state(), lin_w=_lin_w(), **kw)
    batch = {"image": torch.from_numpy(rec["image_0"]), "slice_between": torch.from_numpy(rec["between_0"])}
    if "alpha_from" in rec:
        batch["alpha_from"], batch["alpha_to"] = torch.from_numpy(rec["alpha_from"]), torch.from_numpy(rec["alpha_to"])
    return rec, make, batch


@pytest.mark.parametrize("tag", ["brain_lpips", "cardiac_percept"])
def test_recording_decisions_changes_nothing(tag):
    """OracleStep.train(route=Routing()) with nothing forced is the plain oracle: same losses, same gradients, bit for bit."""
    from oracle import routing
    rec, make, batch = _fixture_step(tag)
    a, b = make(), make()
    ra = a.train(batch["image"], batch["slice_between"], batch.get("alpha_from"), batch.get("alpha_to"))
    rb = b.train(batch["image"], batch["slice_between"], batch.get("alpha_from"), batch.get("alpha_to"), route=routing.Routing())
    for k in ("loss_ae", "loss_ae_dist", "loss_ae_dist_extra", "loss_latent_1"):
        assert ra[k] == rb[k], k
    for k, p in a.ae.params.items():
        assert torch.equal(p.grad, b.ae.params[k].grad), k
        assert torch.equal(p, b.ae.params[k]), k


def _rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))


def test_percept_fixture_sits_on_the_far_side_of_a_tie():
    """step_k3_cardiac_percept (the reference's AETrainerEndToEnd with --use_percept_loss, fp32 on the CPU): ONE decision of the differentiated
    reconstruction branch -- a VGG relu1_1 input that is 1.7e-7 in exact arithmetic -- is taken the other way than fp64 takes it, which moves every
    parameter gradient by ~1e-3 (worst 2.0e-3, median 8e-4).  Evaluated in fp64 UNDER THE FIXTURE'S OWN DECISIONS the fixture's gradients are
    reproduced to 1.2e-4: the fixture is right for its branch, and a path that lands on the other branch (the HIP path does: it decides as fp64
    does, profiles/r06_routing_report.txt) is not wrong.  tests/test_gpu_step.py builds on exactly this."""
    import routing_util as ru
    from oracle import routing
    rec, make, batch = _fixture_step("cardiac_percept")
    fix = {k[6:]: torch.from_numpy(v) for k, v in rec.items() if k.startswith("grad0/")}
    r32 = routing.Routing()
    make().train(batch["image"], batch["slice_between"], route=r32)
    fix_dec = {name: v[2] for name, v in r32.seen.items()}
    r_own, g_own, _ = ru.oracle64_step(make, batch)
    diffs = routing.differing_decisions(r_own, fix_dec)
    assert 1 <= len(diffs) <= 4, ru.describe(diffs)
    assert all(d["rel"] <= 2e-6 for d in diffs), ru.describe(diffs)                     # ties: within a few fp32 roundings of the layer's values
    assert any(d["name"].startswith("lp_rec/in1/") for d in diffs)                      # ... one of them on the branch that is differentiated
    far = sorted(_rel(fix[k], g_own[k]) for k in fix)
    assert far[-1] > 1e-3 and far[len(far) // 2] > 3e-4                                 # the other branch: every gradient moved
    _, g_same, _ = ru.oracle64_step(make, batch, forced=fix_dec)
    near = sorted(_rel(fix[k], g_same[k]) for k in fix)
    assert near[-1] < 2e-4 and near[len(near) // 2] < 4e-5, near                         # the same branch: 1.2e-4 / 1.6e-5 measured


@pytest.mark.parametrize("tag", ["cardiac_lpips", "cardiac_mse", "cardiac_mse_s3"])
def test_other_fixtures_sit_on_no_tie(tag):
    """... and the other step fixtures decide everywhere as fp64 does: their gradients are comparable at face value."""
    import routing_util as ru
    from oracle import routing
    rec, make, batch = _fixture_step(tag)
    r32 = routing.Routing()
    make().train(batch["image"], batch["slice_between"], batch.get("alpha_from"), batch.get("alpha_to"), route=r32)
    r_own, g_own, _ = ru.oracle64_step(make, batch)
    assert routing.differing_decisions(r_own, {name: v[2] for name, v in r32.seen.items()}) == []
    for k, g in g_own.items():
        assert _rel(torch.from_numpy(rec["grad0/" + k]), g) < 3e-5, k
