"""-m gpu: model selection over checkpoints (evaluate/find_best_model.py mirror): every checkpoint is scored on held-out slices --
sub-sample each volume by downsample_steps, synthesise the slices in between, SSIM / PSNR against the originals -- and the per-epoch
score files of the reference are written.  The scores of one checkpoint are re-derived with the oracle's SSIM / PSNR."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_find_best_val_model(tmp_path):
    from evaluate.common import create_super_volume, determine_last_slice
    from evaluate.find_best_model import adjust_and_center_crop, find_best_val_model, load_model_scores
    from oracle import step_oracle
    from superresolution_aniso_mri_amd import train_aesr
    from superresolution_aniso_mri_amd.kwatsch.get_trainer import get_trainer_dynamic
    out = str(tmp_path / "expers")
    tr = train_aesr.main(["--dataset=ACDC", "--model=ae_combined", "--batch_size=4", "--test_batch_size=4", "--latent=16",
                          "--latent_width=8", "--width=32", "--depth=8", "--downsample_steps=2", "--epochs=1", "--lr=0.001",
                          "--ex_loss_weight1=0.05", "--exper_id=m1", "--output_dir=" + out, "--synthetic", "--iters_per_epoch=3",
                          "--image_mix_loss_func=mse", "--epoch_threshold=0"])
    src = os.path.join(out, "m1")
    from superresolution_aniso_mri_amd.data_synth import synthetic_batch
    for epoch in (2, 3):
        for it in range(4):
            tr.train(synthetic_batch(4, 32, 32, seed=500 + 10 * epoch + it), keep_predictions=False)
        tr.save_models(os.path.join(src, "models", "%d.models" % epoch), epoch)
    g = np.random.RandomState(9)
    yy, xx = np.mgrid[0:40, 0:36] / 40.0
    vols = {}
    for p, z in enumerate((9, 8)):          # 8 slices with downsample_steps 3: one remainder slice that is not scored
        base = [np.exp(-((yy - 0.3 - 0.04 * k) ** 2 + (xx - 0.5) ** 2) / 0.03) for k in range(z)]
        vols[p] = {"image": (np.stack(base) * 0.8 + 0.05 * g.rand(z, 40, 36)).astype(np.float32), "patient_id": "p%d" % p,
                   "spacing": np.array([8.0, 1.4, 1.4])}
    scores = find_best_val_model(vols, src, epoch_range=[1, 2, 3], ps_evaluate=32, downsample_steps=3)
    assert list(scores.keys()) == ["1", "2", "3"] and all(np.isfinite(v).all() and 0.0 < v[2] < 1.5 for v in scores.values())
    assert os.path.isfile(os.path.join(src, "model_perf_1_to_3_axis0.npz")) and os.path.isfile(os.path.join(src, "model_perf_synth_1_to_3_axis0.npz"))
    res, epochs, ssim, psnr, vif = load_model_scores(src)
    assert sorted(epochs.tolist()) == [1, 2, 3] and np.allclose(sorted(ssim), sorted(v[0] for v in scores.values()))
    synth = load_model_scores(src, synthesis=True)
    assert synth is not None and len(synth[1]) == 3
    # re-derive checkpoint 2 by hand: same protocol, oracle SSIM / PSNR (fp64 on the host)
    ev, e_args = get_trainer_dynamic(src_path=src, model_nbr=2, eval_mode=True)
    want_s, want_p = [], []
    for v in vols.values():
        img = adjust_and_center_crop(v["image"], 32)
        hr = create_super_volume(ev, torch.from_numpy(img), alpha_range=np.linspace(0, 1, 4)[1:-1], use_original=False, downsample_steps=3,
                                 generate_inbetween_slices=True)["upsampled_image"].numpy()
        last = determine_last_slice(img.shape[0], 3) + 1
        want_s.append(np.mean([step_oracle.ssim(img[z], hr[z]) for z in range(last)]))
        want_p.append(np.mean([step_oracle.psnr(img[z], hr[z]) for z in range(last)]))
    assert abs(scores["2"][0] - np.mean(want_s)) < 1e-6 and abs(scores["2"][1] - np.mean(want_p)) < 1e-4
    # ... and the VIF column: the oracle's restatement of the reference's uint8 vifp_mscale, mean over the finite slices of each volume
    from oracle import vif_oracle
    want_v = []
    for v in vols.values():
        img = adjust_and_center_crop(v["image"], 32)
        hr = create_super_volume(ev, torch.from_numpy(img), alpha_range=np.linspace(0, 1, 4)[1:-1], use_original=False, downsample_steps=3,
                                 generate_inbetween_slices=True)["upsampled_image"].numpy()
        last = determine_last_slice(img.shape[0], 3) + 1
        want_v.append(vif_oracle.compute_vif_for_batch(img[:last], hr[:last])[0])
    assert abs(scores["2"][2] - np.mean(want_v)) < 1e-9
    assert np.isfinite(vif).all() and np.allclose(sorted(vif), sorted(v[2] for v in scores.values()))
    with pytest.raises(ValueError):
        find_best_val_model(vols, src, epoch_range=[77], ps_evaluate=32, downsample_steps=3)
    # optional LPIPS of the scored slices (create_hr_images(compute_percept_loss=True)); synthetic backbone here
    from evaluate.find_best_model import evaluate_interpolation_performance, get_transforms
    from superresolution_aniso_mri_amd.lpips.perceptual import PerceptualLoss
    crit = PerceptualLoss(model="net-lin", net="vgg", use_gpu=True, gpu_ids=[0], device="cuda", vgg_weights="synthetic-hash")
    r = evaluate_interpolation_performance(ev, e_args, vols, transform=get_transforms(32, to_tensor=False), downsample_steps=3,
                                           compute_percept_loss=True, percept_loss=crit)
    assert len(r["lpips"]) == 2 and all(np.isfinite(v) and v > 0 for v in r["lpips"]) and len(r["ssim"]) == 2


def test_evaluate_image_and_stats():
    """evaluate/evaluate_image.py mirror: per-frame held-out-slice synthesis of a 4-D image, its statistics and the comparison grid."""
    from evaluate.evaluate_image import compute_stats, create_compare_image, evaluate_image
    from oracle import step_oracle
    from superresolution_aniso_mri_amd.kwatsch.get_trainer import get_trainer_dynamic
    from superresolution_aniso_mri_amd.networks.net_config import NetworkConfig
    args = dict(model="ae_combined", dataset="ACDC", device="cuda", lr=1e-4, weight_decay=0.0, epochs=2, ex_loss_weight1=0.05,
                use_percept_loss=False, get_masks=False, use_loss_annealing=False, use_extra_latent_loss=False, epoch_threshold=100,
                ae_class="VanillaACAI", image_mix_loss_func="perceptual", vgg_weights="synthetic-hash", width=32, latent_width=8,
                depth=8, latent=16)
    for k, v in NetworkConfig("ae_combined", dataset="ACDC").architecture.items():
        args.setdefault(k, v)
    torch.manual_seed(2)
    tr = get_trainer_dynamic(args)
    g = np.random.RandomState(1)
    img4d = g.rand(3, 7, 40, 36).astype(np.float32)
    res = evaluate_image(tr, {"image": img4d, "patient_id": "patient007", "spacing": np.array([8.0, 1.4, 1.4])}, eval_patch_size=32,
                         downsample_steps=2)
    assert sorted(res["orig_images"].keys()) == [0, 1, 2] and res["synth_images"][1].shape == (7, 32, 32)
    assert float(res["pred_alphas"][0].min()) == 0.5 and res["orig_images"][2].shape == (7, 32, 32)
    one = evaluate_image(tr, {"image": img4d, "patient_id": "patient007", "spacing": np.array([8.0, 1.4, 1.4])}, frame_id=9,
                         eval_patch_size=32, downsample_steps=2)
    assert list(one["synth_images"].keys()) == [2] and np.array_equal(one["synth_images"][2], res["synth_images"][2])
    ssim, psnr, vif, lp = compute_stats(tr, res["orig_images"][0], res["synth_images"][0], normalize=False, downsample_steps=2)
    keep = [1, 3, 5]            # slices 0, 2, 4, 6 are originals at downsample_steps 2
    assert abs(ssim - np.mean([step_oracle.ssim(res["orig_images"][0][z], res["synth_images"][0][z]) for z in keep])) < 1e-6
    assert abs(psnr - np.mean([step_oracle.psnr(res["orig_images"][0][z], res["synth_images"][0][z]) for z in keep])) < 1e-4
    from oracle import vif_oracle
    assert abs(vif - vif_oracle.compute_vif_for_batch(res["orig_images"][0], res["synth_images"][0], downsample_steps=2)[0]) < 1e-9 and lp > 0
    grid = create_compare_image(res["orig_images"][0], res["synth_images"][0], downsample_steps=2)
    assert grid.shape == (1, 7 * (32 + 2) + 2, 3 * (32 + 2) + 2)
