"""-m gpu: the ACAI trainer (kwatsch/trainer_acai.py: auto-encoder + critic, models ``acai`` / ``acai_combined``) on the HIP engine
against vectors produced around the reference's own VanillaACAI / Discriminator modules (tests/golden/step_acai*.npz): losses of
every step, reconstruction / mixes of the first step (rel-L2 1e-5), first-step gradients of BOTH networks (rel-L2 2e-4; the two
backward calls of the reference are one backward of the summed loss here), parameters after the steps (Adam sign-noise bound);
plus the critic module, the checkpoint layout and the plugin path through get_trainer_dynamic."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def make_trainer(tag, rec, lr=1e-3):
    from superresolution_aniso_mri_amd.kwatsch.get_trainer import get_trainer_dynamic
    from superresolution_aniso_mri_amd.networks.net_config import NetworkConfig
    args = dict(model=tag, dataset="OASIS", device="cuda", lr=lr, weight_decay=0.0, epochs=10, width=32, latent_width=8, depth=8,
                latent=16, ex_loss_weight1=0.05, lamb_reg_acai=0.5, use_percept_loss=False, get_masks=False, use_loss_annealing=False,
                use_extra_latent_loss=False, epoch_threshold=100, ae_class="VanillaACAI", image_mix_loss_func="mse")
    for k, v in NetworkConfig(tag, dataset="OASIS").architecture.items():
        args.setdefault(k, v)
    tr = get_trainer_dynamic(args)
    assert type(tr).__name__ == "ACAITrainer" and args["module_trainer_path"] == "kwatsch/trainer_acai.py"
    tr.model.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in rec.items() if k.startswith("p0/")})
    tr.disc_model.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in rec.items() if k.startswith("d0/")})
    return tr


@pytest.mark.parametrize("tag", ["acai_combined", "acai"])
def test_acai_steps_vs_reference_modules(tag):
    rec = dict(np.load(os.path.join(GOLDEN, "step_%s.npz" % tag)))
    tr = make_trainer(tag, rec)
    assert tr.train_combined == (tag == "acai_combined")
    n = len(rec["losses"])
    for step in range(n):
        batch = {"image": torch.from_numpy(rec["image_%d" % step]), "slice_between": torch.from_numpy(rec["between_%d" % step]),
                 "alpha_from": torch.from_numpy(rec["alpha_from"]), "alpha_to": torch.from_numpy(rec["alpha_to"])}
        torch.manual_seed(100 + step)             # the trainer draws alpha = torch.rand(B,1,1,1)/2 exactly where the reference does
        tr.train(batch, keep_predictions=(step == 0))
        want = rec["losses"][step]                # loss_ae, loss_disc, loss_ae_dist, loss_extra, loss_latent
        tol = 2e-5 if step == 0 else 5e-3
        assert abs(tr.losses["loss_ae"][-1] - want[0]) <= tol * abs(want[0])
        assert abs(tr.losses["loss_disc"][-1] - want[1]) <= tol * abs(want[1])
        assert abs(tr.losses["loss_ae_dist"][-1] - want[2]) <= tol * abs(want[2])
        extra_log = tr.losses if tag == "acai_combined" else tr.losses_test
        assert abs(extra_log["loss_ae_dist_extra"][-1] - want[3]) <= tol * abs(want[3])
        assert abs(tr.losses["loss_latent_1"][-1] - want[4]) <= tol * abs(want[4])
        if step == 0:
            assert rel_l2(tr.train_predictions["reconstruction"], rec["out_0"]) < 1e-5
            assert rel_l2(tr.train_predictions["slice_inbetween_mix"], rec["s_mix_0"]) < 1e-5
            for k, p in tr.model.named_parameters():
                assert rel_l2(p.grad, rec["grad0/" + k]) < 2e-4, k
            for k, p in tr.disc_model.named_parameters():
                assert rel_l2(p.grad, rec["dgrad0/" + k]) < 2e-4, k
    for prefix, net in (("p1/", tr.model), ("d1/", tr.disc_model)):
        sd = net.state_dict()
        for k, v in rec.items():
            if not k.startswith(prefix):
                continue
            a, b = sd[k[3:]].double().cpu().numpy(), v.astype(np.float64)
            if "num_batches" in k:
                assert int(a) == int(b), k
                continue
            diff = np.abs(a - b)
            assert diff.max() <= n * 2 * 1e-3 + 1e-6, k
            assert (diff > 2e-4 + 1e-3 * np.abs(b)).mean() <= 0.03, k


def test_discriminator_module_and_checkpoint(tmp_path):
    from superresolution_aniso_mri_amd.networks.acai_vanilla import Discriminator
    rec = dict(np.load(os.path.join(GOLDEN, "step_acai_combined.npz")))
    tr = make_trainer("acai_combined", rec)
    assert isinstance(tr.disc_model, Discriminator) and tr.disc_model.use_sigmoid is False
    keys = list(tr.disc_model.state_dict().keys())
    assert keys[0] == "encoder.0.weight" and set(k[3:] for k in rec if k.startswith("d0/")) == set(keys)
    # eval-mode critic: per-image mean of the encoder output
    tr.disc_model.eval()
    x = torch.rand(5, 1, 32, 32).cuda()
    with torch.no_grad():
        d = tr.disc_model(x)
        feat = tr.disc_model._pass("encoder", [x])[0]
    assert d.shape == (5,) and rel_l2(d, feat.reshape(5, -1).mean(-1)) < 1e-6
    tr.disc_model.train()
    f = str(tmp_path / "3.models")
    tr.save_models(f, 3)
    ck = torch.load(f, map_location="cpu")
    assert set(ck.keys()) == {"model_dict_ae", "optimizer_dict_ae", "model_disc", "optimizer_disc", "epoch"} and ck["epoch"] == 3
    assert "encoder.5.running_mean" in ck["model_disc"]


def test_acai_cli_train_and_reload(tmp_path):
    """train_aesr --model acai_combined on synthetic brain-style triplets (per-sample alphas): settings.yaml + .models with the critic,
    reloaded through get_trainer_dynamic(src_path=..., model_nbr=...) as the evaluation scripts do."""
    from superresolution_aniso_mri_amd import train_aesr
    from superresolution_aniso_mri_amd.kwatsch.get_trainer import get_trainer_dynamic
    out = str(tmp_path / "expers")
    tr = train_aesr.main(["--dataset=OASIS", "--model=acai_combined", "--batch_size=4", "--test_batch_size=4", "--latent=16",
                          "--latent_width=8", "--width=32", "--depth=8", "--downsample_steps=2", "--epochs=1", "--lr=0.001",
                          "--ex_loss_weight1=0.05", "--exper_id=a1", "--output_dir=" + out, "--synthetic", "--iters_per_epoch=3",
                          "--image_mix_loss_func=mse", "--epoch_threshold=0"], brain=True)
    assert type(tr).__name__ == "ACAITrainer" and tr.iters == 4 and len(tr.losses["loss_disc"]) + len(tr.mean_losses["loss_disc"]) >= 1
    src = os.path.join(out, "a1")
    ck = torch.load(os.path.join(src, "models", "1.models"), map_location="cpu")
    assert "model_disc" in ck and "optimizer_disc" in ck
    ev, args = get_trainer_dynamic(src_path=src, model_nbr=1, eval_mode=True)
    assert type(ev).__name__ == "ACAITrainer" and args["trainer_class"] == "ACAITrainer"
    x = torch.rand(3, 1, 32, 32)
    assert rel_l2(ev.predict(x), tr.predict(x)) < 1e-6
