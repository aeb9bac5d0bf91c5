#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REFERENCE's own modules on CPU (build container only).

TEST INFRASTRUCTURE.  The reference (/root/reference) has no tests or golden vectors
(SURVEY.md section 4), so the oracle is pinned against the reference modules themselves: this script
imports them (with `sys.modules` stubs for the missing third-party imports they never use on
this path: torchvision, skimage) and stores inputs + expected outputs as small fixtures.  Only
data is written; no reference source is copied.  The reference trainer classes cannot be
constructed without CUDA (kwatsch/trainer_ae.py:51), so the train step is driven by the
arithmetic of AETrainerEndToEnd.train restated here around the *reference* network modules.

Run:  python oracle/make_golden.py          (needs /root/reference; not available on the GPU box)
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("AESR_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
from oracle import lpips_oracle, step_oracle  # noqa: E402  (input generators + synthetic backbone spec)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _fake_vgg16(pretrained=False, **kw):
    """Stand-in for torchvision.models.vgg16: the public VGG16-D `features` layout filled with the
    deterministic synthetic weights (the ImageNet weights are a network download, unavailable)."""
    layers, cin = [], 3
    for v in lpips_oracle.VGG16_CFG:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v, 3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    feats = nn.Sequential(*layers)
    feats.load_state_dict({k.replace("features.", ""): v for k, v in lpips_oracle.hash_vgg16_state().items()})
    return types.SimpleNamespace(features=feats)


def import_reference():
    tv = _stub("torchvision")
    tv.transforms = _stub("torchvision.transforms")
    tv.utils = _stub("torchvision.utils", make_grid=None)
    tv.models = _stub("torchvision.models", vgg16=_fake_vgg16)
    _stub("skimage")
    _stub("skimage.metrics", structural_similarity=None, peak_signal_noise_ratio=None)
    _stub("skimage.measure")
    sys.path.insert(0, REF)
    import networks.acai_vanilla as av
    import networks.acai_vanilla_strided as avs
    import networks.acai_vanilla_modified as avm
    import lpips.networks_basic as nb
    return av, avs, avm, nb


def np_state(sd):
    return {k: v.detach().cpu().numpy().copy() for k, v in sd.items()}


def small_args(depth=8, latent=16, width=32, latent_width=8):
    return dict(width=width, latent_width=latent_width, depth=depth, latent=latent, colors=1, device="cpu",
                use_batchnorm=True, use_sigmoid=True, n_res_block=None)


def gen_ae_small(av, avs, avm):
    """fwd + bwd of one train-mode pass for each AE variant at odd/even sizes."""
    classes = {"VanillaACAI": av.VanillaACAI, "VanillaACAIStrided": avs.VanillaACAIStrided, "LargerAE": avm.LargerAE}
    for cname, cls in classes.items():
        for (N, H, W) in ((2, 28, 28), (4, 33, 33), (2, 40, 36)):
            torch.manual_seed(1000 + H)
            model = cls(small_args())
            model.train()
            sd0 = np_state(model.state_dict())
            x = torch.rand(N, 1, H, W)
            x.requires_grad_(True)
            z = model.encode(x)
            out = model.decode(z)
            # the decoder's output size differs from the input for odd sizes -> loss against a fixed target
            tgt = torch.rand(out.shape, generator=torch.Generator().manual_seed(7))
            loss = F.mse_loss(out, tgt) + 0.1 * (z ** 2).mean()
            loss.backward()
            rec = {"x": x.detach().numpy(), "tgt": tgt.numpy(), "z": z.detach().numpy(), "out": out.detach().numpy(),
                   "loss": np.float64(loss.item()), "dx": x.grad.numpy()}
            rec.update({"p0/" + k: v for k, v in sd0.items()})
            rec.update({"grad/" + k: p.grad.numpy() for k, p in model.named_parameters()})
            rec.update({"p1/" + k: v for k, v in np_state(model.state_dict()).items() if "running" in k or "num_b" in k})
            model.eval()
            with torch.no_grad():
                rec["out_eval"] = model(x.detach()).numpy()
            np.savez_compressed(os.path.join(OUT, "ae_small_%s_%dx%dx%d.npz" % (cname, N, H, W)), **rec)


def gen_ae_init(av):
    """RNG-exact init: seed -> parameter statistics of the full ACDC model (443 777 params)."""
    torch.manual_seed(892372)
    m = av.VanillaACAI(dict(width=128, latent_width=32, depth=32, latent=128, colors=1, device="cpu",
                            use_batchnorm=True, use_sigmoid=True, n_res_block=None))
    rec = {"nparams": np.int64(sum(p.numel() for p in m.parameters()))}
    for k, p in m.named_parameters():
        rec["sum/" + k] = np.float64(p.double().sum().item())
        rec["abs/" + k] = np.float64(p.double().abs().sum().item())
        rec["head/" + k] = p.detach().flatten()[:4].numpy()
    np.savez_compressed(os.path.join(OUT, "ae_init_acdc.npz"), **rec)
    return m


def gen_ae_acdc_probe(m):
    """Full-size C2/C3 model, 1 triplet at 160x160: sampled outputs / grads + norms."""
    image, between = step_oracle.synthetic_triplets(1, 160, 160, seed=892372)
    m.train()
    z = m.encode(image)
    out = m.decode(z)
    loss = F.mse_loss(out, image)
    loss.backward()
    idx = np.random.RandomState(0).randint(0, out.numel(), size=64)
    zidx = np.random.RandomState(1).randint(0, z.numel(), size=64)
    rec = dict(loss=np.float64(loss.item()), out_idx=idx, out_val=out.detach().flatten()[idx].numpy(),
               z_idx=zidx, z_val=z.detach().flatten()[zidx].numpy(),
               out_norm=np.float64(out.double().norm().item()), z_norm=np.float64(z.double().norm().item()))
    for k, p in m.named_parameters():
        rec["gnorm/" + k] = np.float64(p.grad.double().norm().item())
        rec["ghead/" + k] = p.grad.flatten()[:4].numpy()
    for k, b in m.named_buffers():
        if "running" in k:
            rec["bn/" + k] = b.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "ae_acdc_probe.npz"), **rec)


def build_pnetlin(nb):
    net = nb.PNetLin(pnet_type="vgg", pnet_rand=True, pnet_tune=False, use_dropout=True, spatial=False,
                     version="0.1", lpips=True)
    lin_sd = torch.load(os.path.join(REF, "lpips", "weights", "v0.1", "vgg.pth"), map_location="cpu")
    net.load_state_dict(lin_sd, strict=False)
    net.eval()
    return net, lin_sd


def gen_lpips(nb):
    net, lin_sd = build_pnetlin(nb)
    # ship the lin-layer weights (data) with the product: LPIPS v0.1 linear calibration for VGG
    wdir = os.path.join(ROOT, "superresolution_aniso_mri_amd", "lpips", "weights", "v0.1")
    os.makedirs(wdir, exist_ok=True)
    np.savez(os.path.join(wdir, "vgg_lin.npz"),
             **{"lin%d" % k: lin_sd["lin%d.model.1.weight" % k].numpy().reshape(-1) for k in range(5)})
    # (a) head only: synthetic tap features -> per-layer + total distance and grads wrt branch-1 features
    g = torch.Generator().manual_seed(11)
    shapes = [(2, 64, 12, 12), (2, 128, 6, 6), (2, 256, 3, 3), (2, 512, 2, 2), (2, 512, 1, 1)]
    f0 = [torch.rand(s, generator=g) for s in shapes]
    f1 = [torch.rand(s, generator=g).requires_grad_(True) for s in shapes]
    import lpips.common as util
    res = []
    for k in range(5):
        d = (util.normalize_tensor(f0[k]) - util.normalize_tensor(f1[k])) ** 2
        res.append(nb.spatial_average(net.lins[k].model(d), keepdim=True))
    val = sum(res)
    val.sum().backward()
    rec = {"val": val.detach().numpy()}
    for k in range(5):
        rec["f0_%d" % k], rec["f1_%d" % k] = f0[k].numpy(), f1[k].detach().numpy()
        rec["res_%d" % k], rec["g1_%d" % k] = res[k].detach().numpy(), f1[k].grad.numpy()
    np.savez_compressed(os.path.join(OUT, "lpips_head.npz"), **rec)
    # (b) full path on 1-channel images (broadcast in ScalingLayer), incl. the perceptual.py 2x-1 scaling
    for (N, H, W) in ((2, 32, 32), (1, 48, 40)):
        g = torch.Generator().manual_seed(100 + H)
        ref = torch.rand(N, 1, H, W, generator=g)
        syn = (ref + 0.1 * torch.randn(N, 1, H, W, generator=g)).clamp(0, 1).requires_grad_(True)
        # trainer call: percept_criterion(reference, synthesized, normalize=True) -> PerceptualLoss.forward(
        # pred=reference, target=synthesized) -> model.forward(target, pred) = PNetLin(in0=2*syn-1, in1=2*ref-1)
        in0, in1 = 2 * syn - 1, 2 * ref - 1
        d = net.forward(in0, in1)
        d.mean().backward()
        taps = net.net.forward(net.scaling_layer(in0))
        rec = dict(ref=ref.numpy(), syn=syn.detach().numpy(), d=d.detach().numpy(), dsyn=syn.grad.numpy())
        for k, t in enumerate(taps):
            rec["tap%d_norm" % k] = np.float64(t.double().norm().item())
            rec["tap%d_head" % k] = t.detach().flatten()[:8].numpy()
        np.savez_compressed(os.path.join(OUT, "lpips_full_%dx%dx%d.npz" % (N, H, W)), **rec)
    return net


def gen_steps(av, net):
    """k=3 ae_combined train steps (cardiac 0.5/0.5 lerp and brain per-sample alphas; LPIPS and MSE mix loss)."""
    for tag, mix, brain in (("cardiac_lpips", "perceptual", False), ("brain_lpips", "perceptual", True),
                            ("cardiac_mse", "mse", False)):
        torch.manual_seed(4242)
        model = av.VanillaACAI(small_args())
        sd0 = np_state(model.state_dict())
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=0.0, betas=(0.9, 0.999))
        lam = 0.05
        B, H, W = 3, 32, 32
        rec = {"p0/" + k: v for k, v in sd0.items()}
        losses = []
        for step in range(3):
            image, between = step_oracle.synthetic_triplets(B, H, W, seed=500 + step)
            rec["image_%d" % step], rec["between_%d" % step] = image.numpy(), between.numpy()
            model.train()
            z = model.encode(image)
            out = model.decode(z)
            loss_ae = F.mse_loss(out, image, reduction="mean")
            if brain:
                af = torch.tensor([[0.25], [0.5], [0.75]])
                at = 1 - af
                rec["alpha_from"], rec["alpha_to"] = af.numpy(), at.numpy()
                z_mix = af[:, :, None, None] * z[:B] + at[:, :, None, None] * z[B:]
            else:
                a05 = torch.tensor([0.5])[:, None, None, None]
                z_mix = a05 * z[:B] + (1 - a05) * z[B:]
            s_mix = model.decode(z_mix)
            z_ref = model.encode(between)
            lat = F.mse_loss(z_mix, z_ref)
            if mix == "perceptual":
                # PerceptualLoss.forward(pred=between, target=s_mix, normalize=True) -> net(2*s_mix-1, 2*between-1)
                extra = net.forward(2 * s_mix - 1, 2 * between - 1).mean()
            else:
                extra = F.mse_loss(between, s_mix)
            loss = loss_ae + lam * extra
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append([loss.item(), loss_ae.item(), (lam * extra).item(), lat.item()])
            if step == 0:
                rec["z_0"], rec["out_0"], rec["s_mix_0"] = z.detach().numpy(), out.detach().numpy(), s_mix.detach().numpy()
                rec.update({"grad0/" + k: p.grad.numpy().copy() for k, p in model.named_parameters()})
        rec["losses"] = np.array(losses, dtype=np.float64)
        rec.update({"p3/" + k: v for k, v in np_state(model.state_dict()).items()})
        np.savez_compressed(os.path.join(OUT, "step_k3_%s.npz" % tag), **rec)


# ---- the reference's OWN trainer classes on the CPU ---------------------------------------------------------------------------
class _DummyMeta(type):
    def __getattr__(cls, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return _Dummy


class _Dummy(metaclass=_DummyMeta):
    """Stand-in for names of third-party packages the trainer modules import at module level and never call on this path."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return None


class _AnyModule(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return _Dummy


def _any_stub(name):
    m = _AnyModule(name)
    m.__path__ = []
    sys.modules[name] = m
    return m


def import_reference_trainers():
    """kwatsch/cardiac/trainer_ae.py, kwatsch/brain/trainer_ae.py, kwatsch/trainer_ae.py of the reference, imported as they are.
    Their import chain (kwatsch/base_trainer.py:1-13 -> evaluate/*, datasets/*) pulls in packages that are absent here and that
    the train step never calls (tensorboard, SimpleITK, imageio, cv2, batchgenerators, gpustat, torchvision.datasets): each
    ModuleNotFoundError is answered with an empty stand-in module.  The ONE behavioural shim: ``torch.cuda.FloatTensor`` (used
    once, kwatsch/trainer_ae.py:51, to build the constant 0.5) makes a CPU tensor.  Nothing of the step itself is replaced."""
    import_reference()
    _any_stub("torch.utils.tensorboard")
    _any_stub("torch.utils.tensorboard.writer")
    tv = sys.modules["torchvision"]
    for sub in ("datasets", "transforms", "utils"):
        setattr(tv, sub, _any_stub("torchvision." + sub))
    torch.cuda.FloatTensor = lambda data, device=None: torch.FloatTensor(data)
    for _ in range(64):
        try:
            import kwatsch.trainer_ae as tae
            import kwatsch.cardiac.trainer_ae as cta
            import kwatsch.brain.trainer_ae as bta
            return tae, cta, bta
        except ModuleNotFoundError as e:
            _any_stub(e.name)
    raise RuntimeError("could not import the reference trainers")


def trainer_args(mix="mse", lr=1e-3, epochs=10, **over):
    a = small_args()
    a.update(model="ae_combined", lr=lr, weight_decay=0.0, epochs=epochs, use_percept_loss=False, get_masks=False,
             use_loss_annealing=False, use_extra_latent_loss=False, ex_loss_weight1=0.05, image_mix_loss_func=mix,
             epoch_threshold=100)
    a.update(over)
    return a


def _run_reference_trainer(av, cls, args, batches, seed=4242, epochs_at=None):
    """Construct the reference trainer ``cls`` around the reference ``VanillaACAI`` and call its real ``train()`` once per batch.
    Returns the record: initial / final parameters, the scalars the trainer logged, first-step predictions and gradients."""
    torch.manual_seed(seed)
    model = av.VanillaACAI(args)
    rec = {"p0/" + k: v for k, v in np_state(model.state_dict()).items()}
    trainer = cls(args, model)
    z_seen = []
    hook = model.enc.register_forward_hook(lambda m, i, o: z_seen.append(o.detach().clone()))
    keys = None
    rows = []
    for step, batch in enumerate(batches):
        if epochs_at is not None:
            trainer.epoch = int(epochs_at[step])
        n0 = {k: len(v) for k, v in trainer.losses.items()}
        z_seen.clear()
        trainer.train(dict(batch), keep_predictions=True)
        if keys is None:
            keys = [k for k in ("loss_ae", "loss_ae_dist", "loss_ae_dist_extra", "loss_latent_1") if k in trainer.losses]
        for k in trainer.losses:
            assert len(trainer.losses[k]) == n0.get(k, 0) + 1, "one value per key per step expected (%s)" % k
        rows.append([trainer.losses[k][-1] for k in keys])
        rec["image_%d" % step], rec["between_%d" % step] = batch["image"].numpy(), batch["slice_between"].numpy()
        if step == 0:
            pr = trainer.train_predictions
            rec["z_0"] = z_seen[0].numpy()                      # first encoder call of the step = enc(x)
            rec["out_0"], rec["s_mix_0"], rec["z_mix_0"] = pr["reconstruction"].numpy(), pr["slice_inbetween_mix"].numpy(), pr["z_mix"].numpy()
            rec.update({"grad0/" + k: p.grad.numpy().copy() for k, p in model.named_parameters()})
    hook.remove()
    rec["losses"] = np.array(rows, dtype=np.float64)
    rec["loss_keys"] = np.array(keys)
    rec["iters"] = np.array(trainer.iters)
    rec["loss_weights"] = np.asarray(trainer.loss_weights, dtype=np.float64)
    nsteps = len(batches)
    rec.update({"p%d/" % nsteps + k: v for k, v in np_state(model.state_dict()).items()})
    return rec, trainer


def _triplet_batches(n, B, H, W, seed0, brain=False):
    out = []
    for step in range(n):
        image, between = step_oracle.synthetic_triplets(B, H, W, seed=seed0 + step)
        b = {"image": image, "slice_between": between}
        if brain:
            af = torch.tensor([[0.25], [0.5], [0.75]])[:B] if B <= 3 else torch.linspace(0.2, 0.8, B)[:, None]
            b["alpha_from"], b["alpha_to"] = af, 1 - af
        out.append(b)
    return out


def gen_trainer_steps(av):
    """tests/golden/step_k3_*.npz: three optimisation steps made by the reference's own ``AETrainerEndToEnd.train``
    (kwatsch/cardiac/trainer_ae.py:10-50), ``AETrainerExtension1Brain.train`` (kwatsch/brain/trainer_ae.py:92-132) and the plain
    ``AEBaseTrainer.train`` (kwatsch/trainer_ae.py:71-109) on the CPU -- pass order, BatchNorm update order, loss composition and
    logging are the reference's, not a restatement.  LPIPS uses the reference's PerceptualLoss -> DistModel -> PNetLin with the
    local lin weights and the synthetic backbone (the ImageNet weights are a download)."""
    tae, cta, bta = import_reference_trainers()
    cases = [("cardiac_mse", cta.AETrainerEndToEnd, trainer_args("mse"), False),
             ("cardiac_lpips", cta.AETrainerEndToEnd, trainer_args("perceptual"), False),
             ("brain_lpips", bta.AETrainerExtension1Brain, trainer_args("perceptual"), True),
             # --use_percept_loss: LPIPS as the reconstruction loss too (kwatsch/base_trainer.py:165-175), synthesis loss follows it
             ("cardiac_percept", cta.AETrainerEndToEnd, trainer_args(None, use_percept_loss=True), False),
             # plain `ae` (reconstruction loss only; latent loss and the 0.5 mix are logged)
             ("ae_plain", tae.AEBaseTrainer, trainer_args("mse", model="ae"), False),
             # three pooling stages (README-literal latent_width: width // latent_width = 8 -> scales 3)
             ("cardiac_mse_s3", cta.AETrainerEndToEnd, trainer_args("mse", width=32, latent_width=4), False),
             # the reference's default learning rate: parameters after 3 steps agree far inside Adam's sign-noise band
             ("cardiac_mse_lr1e-5", cta.AETrainerEndToEnd, trainer_args("mse", lr=1e-5), False)]
    for tag, cls, args, brain in cases:
        if args.get("image_mix_loss_func") is None:
            args.pop("image_mix_loss_func")
        rec, tr = _run_reference_trainer(av, cls, args, _triplet_batches(3, 3, 32, 32, 500, brain))
        if brain:
            rec["alpha_from"], rec["alpha_to"] = np.array([[0.25], [0.5], [0.75]], np.float32), np.array([[0.75], [0.5], [0.25]], np.float32)
        rec["trainer_class"] = np.array(cls.__name__)
        np.savez_compressed(os.path.join(OUT, "step_k3_%s.npz" % tag), **rec)
        print("  step_k3_%s: %s keys %s, losses[0] %s" % (tag, cls.__name__, list(rec["loss_keys"]), rec["losses"][0]))
    # loss annealing (kwatsch/base_trainer.py:456-459, kwatsch/cardiac/trainer_ae.py:80-83): 4 epochs, one step in each
    args = trainer_args("mse", epochs=4, use_loss_annealing=True)
    rec, tr = _run_reference_trainer(av, cta.AETrainerEndToEnd, args, _triplet_batches(4, 3, 32, 32, 700), epochs_at=[0, 1, 2, 3])
    rec["trainer_class"] = np.array("AETrainerEndToEnd")
    np.savez_compressed(os.path.join(OUT, "step_k4_cardiac_anneal.npz"), **rec)
    # BASELINE configs[1] / [2] at their own size (12 triplets of 160x160, depth 32, latent 128), ONE step of the reference trainer:
    # stored as probes (scalars, norms, 64 sampled values per tensor) -- the batch is re-made from its seed by the tests
    for tag, mix in (("c2", "mse"), ("c3", "perceptual")):
        args = trainer_args(mix, lr=1e-5, width=128, latent_width=32, depth=32, latent=128)
        torch.manual_seed(892372)
        model = av.VanillaACAI(args)
        from superresolution_aniso_mri_amd.data_synth import synthetic_batch
        batch = synthetic_batch(12, 160, 160, seed=892372)
        trainer = cta.AETrainerEndToEnd(args, model)
        init_sum = {k: float(v.double().sum()) for k, v in model.state_dict().items()}
        trainer.train({"image": batch["image"], "slice_between": batch["slice_between"]}, keep_predictions=True)
        pr = trainer.train_predictions
        idx = torch.randperm(pr["reconstruction"].numel(), generator=torch.Generator().manual_seed(1))[:64]
        idx_s = torch.randperm(pr["slice_inbetween_mix"].numel(), generator=torch.Generator().manual_seed(2))[:64]
        rec = {"losses": np.array([trainer.losses[k][-1] for k in ("loss_ae", "loss_ae_dist", "loss_ae_dist_extra", "loss_latent_1")]),
               "out_idx": idx.numpy(), "out_val": pr["reconstruction"].flatten()[idx].numpy(), "out_norm": np.float64(pr["reconstruction"].double().norm()),
               "s_idx": idx_s.numpy(), "s_val": pr["slice_inbetween_mix"].flatten()[idx_s].numpy(), "s_norm": np.float64(pr["slice_inbetween_mix"].double().norm()),
               "zmix_norm": np.float64(pr["z_mix"].double().norm())}
        for k, p_ in model.named_parameters():
            rec["gnorm/" + k] = np.float64(p_.grad.double().norm())
            rec["ghead/" + k] = p_.grad.flatten()[:8].numpy().copy()
        for k, v in model.state_dict().items():
            if "running" in k:
                rec["bn/" + k] = v.numpy().copy()
            rec["init_sum/" + k] = np.float64(init_sum[k])
        np.savez_compressed(os.path.join(OUT, "step_probe_%s.npz" % tag), **rec)
        print("  step_probe_%s: losses %s" % (tag, rec["losses"]))


def gen_val_volumes(av):
    """kwatsch/base_trainer.py:67-99,149-162: ``validate(validation_batch, image_dict=...)`` of the reference's OWN ``AETrainerEndToEnd`` on the
    CPU -- ``_generate_val_volumes`` -> ``evaluate.evaluate_image.evaluate_image`` (:37-80: AdjustToPatchSize + CenterCrop, every 2nd slice
    kept, the held-out ones synthesised at alpha 0.5, all slices reconstructed) -> ``create_compare_image`` (:83-106) per in-memory 4-D
    patient.  Placement shims only: ``Tensor.to('cuda')`` (evaluate/common.py:185, kwatsch/base_trainer.py:317) is the identity here and
    ``latent_space_interp``'s default device is bound to 'cpu'.  torchvision is absent: ``transforms.Compose`` is its published definition
    (apply in order) and ``make_grid`` RECORDS its argument -- the fixture stores the tensor the reference hands to ``make_grid`` with its
    nrow / padding / pad_value, so the grid layout itself stays pinned by the layout tests of the build's own ``make_grid``."""
    import functools
    tae, cta, bta = import_reference_trainers()
    import evaluate.common as ec
    import evaluate.evaluate_image as ei
    ec.latent_space_interp = functools.partial(ec.latent_space_interp, device="cpu")

    class Compose(object):
        def __init__(self, ts):
            self.transforms = ts

        def __call__(self, x):
            for t in self.transforms:
                x = t(x)
            return x

    ei.transforms.Compose = Compose
    seen = []

    def recording_make_grid(t, nrow=8, padding=2, normalize=False, pad_value=0):
        seen.append((t.detach().clone(), int(nrow), int(padding), float(pad_value), bool(normalize)))
        return t

    ei.make_grid = recording_make_grid
    if not hasattr(np, "bool"):
        np.bool = bool                              # create_compare_image uses the alias numpy removed in 1.24 (evaluate_image.py:93-94)
    to0 = torch.Tensor.to

    def to_cpu_if_cuda(self, *a, **k):
        if a and isinstance(a[0], str) and a[0].startswith("cuda"):
            return self
        return to0(self, *a, **k)

    args = trainer_args("mse", lr=1e-3)
    rec, trainer = _run_reference_trainer(av, cta.AETrainerEndToEnd, args, _triplet_batches(2, 3, 32, 32, 900))
    rec = {"p/" + k: v for k, v in np_state(trainer.model.state_dict()).items()}
    g = np.random.RandomState(31)

    def patient(t, z, h, w):
        low = g.rand(t, z, (h + 7) // 8, (w + 7) // 8)
        return np.clip(np.kron(low, np.ones((1, 1, 8, 8)))[:, :, :h, :w] * 0.8 + 0.15 * g.rand(t, z, h, w), 0, 1).astype(np.float32)

    image_dict = {3: {"image": patient(3, 7, 40, 36), "patient_id": "patient003", "spacing": np.array([8.0, 1.4, 1.4])},
                  17: {"image": patient(2, 8, 30, 44), "patient_id": "patient017", "spacing": np.array([10.0, 1.25, 1.25])}}
    val = _triplet_batches(1, 4, 32, 32, 950)[0]
    frame_id = 1
    torch.Tensor.to = to_cpu_if_cuda
    try:
        out = trainer.validate(dict(val), image_dict=image_dict, frame_id=frame_id, generate_images=False)
        ev = {p: ei.evaluate_image(trainer, image_dict[p], frame_id=frame_id, downsample_steps=2, eval_patch_size=args["width"]) for p in image_dict}
    finally:
        torch.Tensor.to = to0
    assert sorted(out["synthesized_vols"].keys()) == [3, 17] and len(seen) >= 2
    rec["val/image"], rec["val/slice_between"] = val["image"].numpy(), val["slice_between"].numpy()
    rec["val/loss_ae"] = np.float64(out["loss_ae"])
    rec["frame_id"], rec["patients"] = np.array(frame_id), np.array([3, 17])
    for i, p in enumerate((3, 17)):
        t, nrow, padding, pad_value, normalize = seen[i]
        assert not normalize and np.array_equal(out["synthesized_vols"][p], t.numpy())      # the recorder returned its argument
        rec["p%d/image4d" % p] = image_dict[p]["image"]
        rec["p%d/grid_input" % p] = t.numpy()
        rec["p%d/grid_args" % p] = np.array([nrow, padding, pad_value])
        rec["p%d/alphas" % p] = np.asarray(out["alphas"][p], dtype=np.float32)
        rec["p%d/orig" % p] = ev[p]["orig_images"][frame_id]
        rec["p%d/synth" % p] = ev[p]["synth_images"][frame_id]
    np.savez_compressed(os.path.join(OUT, "val_volumes.npz"), **rec)
    print("  val_volumes: loss_ae %.6f, grid inputs %s" % (out["loss_ae"], [tuple(seen[i][0].shape) for i in range(2)]))


def gen_supervolume(av):
    """generate_hr_volumes.py:12-101 arithmetic (z=5, n=3) around the reference model in eval mode; the
    reference function itself hard-codes .to('cuda') so its loop is restated here verbatim in meaning:
    per alpha re-encode both neighbour stacks, lerp, decode, interleave, clamp."""
    torch.manual_seed(77)
    model = av.VanillaACAI(small_args())
    # make running stats non-trivial
    model.train()
    with torch.no_grad():
        model(torch.rand(4, 1, 32, 32))
    model.eval()
    vol = torch.rand(5, 1, 32, 32, generator=torch.Generator().manual_seed(3)) * 1.2 - 0.1
    alpha_range = np.linspace(0, 1, 3 + 2, endpoint=True)[1:-1]
    images2, images1 = vol[1:], vol[:-1]
    interp = None
    with torch.no_grad():
        for alpha in alpha_range:
            l1, l2 = model.encode(images2.float()), model.encode(images1.float())
            inter = model.decode(alpha * l1 + (1 - alpha) * l2)
            interp = inter if interp is None else torch.cat([interp, inter], dim=1)
        new = None
        for i in range(vol.shape[0] - 1):
            new = torch.cat([vol[i], interp[i]]) if new is None else torch.cat([new, vol[i], interp[i]], dim=0)
        new = torch.clamp(torch.cat([new, vol[i + 1]]), min=0, max=1.)
    rec = {"vol": vol.numpy(), "alpha_range": alpha_range, "hr": new.numpy()}
    rec.update({"p/" + k: v for k, v in np_state(model.state_dict()).items()})
    np.savez_compressed(os.path.join(OUT, "supervolume.npz"), **rec)


def gen_supervolume_eval(av):
    """evaluate/common.py:134-235 ``create_super_volume`` -- the evaluation protocol (keep every ``downsample_steps``-th slice,
    synthesise the ones in between, append the remainder slices) -- run as the reference's OWN function on the CPU: its
    ``latent_space_interp`` default device ('cuda') is bound to 'cpu', nothing else is touched.  SimpleITK / imageio are
    import-time only for this function: stubbed."""
    import functools
    for name in ("SimpleITK", "imageio"):
        if name not in sys.modules:
            _stub(name, sitkLanczosWindowedSinc=0, Image=object, imsave=None)
    import evaluate.common as ec
    ec.latent_space_interp = functools.partial(ec.latent_space_interp, device="cpu")

    class _Trainer(object):                 # the three methods the function calls (kwatsch/base_trainer.py:216-336 on CPU)
        def __init__(self, model):
            self.model = model

        def encode(self, x, use_sr_model=False, **kw):
            with torch.no_grad():
                return self.model.encode(x.float())

        def decode(self, z, use_sr_model=False, **kw):
            with torch.no_grad():
                return self.model.decode(z)

        def predict(self, x, **kw):
            return self.decode(self.encode(x))

    torch.manual_seed(78)
    model = av.VanillaACAI(small_args())
    model.train()
    with torch.no_grad():
        model(torch.rand(4, 1, 32, 32))
    model.eval()
    tr = _Trainer(model)
    rec = {"p/" + k: v for k, v in np_state(model.state_dict()).items()}
    cases = [("default", 5, dict()),
             ("inbetween_rem", 8, dict(alpha_range=np.linspace(0, 1, 4)[1:-1], generate_inbetween_slices=True)),
             ("inbetween_even", 7, dict(alpha_range=np.array([0.5]), generate_inbetween_slices=True, downsample_steps=2)),
             ("downsample_only", 9, dict(alpha_range=np.array([0.3, 0.6]), downsample_steps=4))]
    for tag, z, kw in cases:
        vol = torch.rand(z, 16, 20, generator=torch.Generator().manual_seed(10 + z)) * 1.2 - 0.1
        out = ec.create_super_volume(tr, vol.clone(), use_original=True, **kw)
        rec[tag + "/vol"] = vol.numpy()
        rec[tag + "/hr"] = out["upsampled_image"].numpy()
        rec[tag + "/pred_alphas_shape"] = np.array(out["pred_alphas"].shape)
        rec[tag + "/pred_alphas_first"] = out["pred_alphas"][:, 0, 0, 0].numpy()
        if "alpha_range" in kw:
            rec[tag + "/alpha_range"] = np.asarray(kw["alpha_range"], dtype=np.float64)
        rec[tag + "/downsample_steps"] = np.array(-1 if kw.get("downsample_steps") is None else kw["downsample_steps"])
        rec[tag + "/generate_inbetween_slices"] = np.array(int(kw.get("generate_inbetween_slices", False)))
    rec["determine_last_slice"] = np.array([[n, d, ec.determine_last_slice(n, d)] for n in (5, 8, 9, 12) for d in (2, 3, 4)])
    np.savez_compressed(os.path.join(OUT, "supervolume_eval.npz"), **rec)


def gen_acai_steps(av):
    """kwatsch/trainer_acai.py:46-127 (ACAITrainer.train / get_loss_disc) restated around the reference's OWN VanillaACAI and
    Discriminator modules (the trainer class itself needs CUDA): two steps of ``acai_combined`` (MSE synthesis loss) and one of
    plain ``acai``; alpha drawn as the reference draws it (torch.rand(B,1,1,1)/2 after torch.manual_seed)."""
    import torch.nn.functional as F
    for tag, combined, nsteps in (("acai_combined", True, 2), ("acai", False, 1)):
        torch.manual_seed(41)
        model = av.VanillaACAI(small_args())
        disc = av.Discriminator(small_args())
        lr, lamb, lam, gamma, B = 1e-3, 0.5, 0.05, 0.2, 3
        opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=0.0, betas=(0.9, 0.999))
        opt_d = torch.optim.Adam(disc.parameters(), lr=lr, weight_decay=0.0, betas=(0.9, 0.999))
        rec = {"p0/" + k: v for k, v in np_state(model.state_dict()).items()}
        rec.update({"d0/" + k: v for k, v in np_state(disc.state_dict()).items()})
        g = torch.Generator().manual_seed(7)
        af = torch.tensor([0.25, 0.5, 0.75])[:, None]
        rec["alpha_from"], rec["alpha_to"] = af.numpy(), (1 - af).numpy()
        losses = []
        model.train()
        for step in range(nsteps):
            x = torch.rand(2 * B, 1, 32, 32, generator=g)
            btw = torch.rand(B, 1, 32, 32, generator=g)
            torch.manual_seed(100 + step)
            alpha = torch.rand(B, 1, 1, 1) / 2
            z = model.encode(x)
            out = model.decode(z)
            loss_ae_dist = F.mse_loss(out, x, reduction="mean")
            loss_disc_l2 = torch.mean(disc(out + gamma * (x - out)) ** 2)
            out_mix = model.decode(alpha * z[:B] + (1 - alpha) * z[B:])
            disc_mix = disc(out_mix)
            loss_ae_l2 = torch.mean(disc_mix ** 2)
            loss_disc_dist = F.mse_loss(disc_mix, alpha.reshape(-1), reduction="mean")
            loss_ae = loss_ae_dist + lamb * loss_ae_l2
            loss_disc = loss_disc_dist + loss_disc_l2
            z_mix = af[:, :, None, None] * z[:B] + (1 - af)[:, :, None, None] * z[B:]
            if combined:
                s_mix = model.decode(z_mix)
                z_ref = model.encode(btw)
                loss_extra = lam * F.mse_loss(btw, s_mix)
                loss_ae = loss_ae + loss_extra
            else:
                model.eval()
                with torch.no_grad():
                    s_mix = model.decode(z_mix)
                    z_ref = model.encode(btw)
                    loss_extra = lam * F.mse_loss(btw, s_mix)
                model.train()
            loss_latent = F.mse_loss(z_mix, z_ref)
            opt.zero_grad()
            opt_d.zero_grad()
            loss_ae.backward(retain_graph=True)
            loss_disc.backward()
            if step == 0:
                rec.update({"grad0/" + k: p.grad.numpy().copy() for k, p in model.named_parameters()})
                rec.update({"dgrad0/" + k: p.grad.numpy().copy() for k, p in disc.named_parameters()})
                rec["out_0"], rec["s_mix_0"], rec["out_mix_0"] = out.detach().numpy(), s_mix.detach().numpy(), out_mix.detach().numpy()
            opt.step()
            opt_d.step()
            rec["image_%d" % step], rec["between_%d" % step], rec["alpha_%d" % step] = x.numpy(), btw.numpy(), alpha.reshape(-1).numpy()
            losses.append([loss_ae.item(), loss_disc.item(), loss_ae_dist.item(), loss_extra.item(), loss_latent.item()])
        rec["losses"] = np.array(losses, dtype=np.float64)
        rec.update({"p1/" + k: v for k, v in np_state(model.state_dict()).items()})
        rec.update({"d1/" + k: v for k, v in np_state(disc.state_dict()).items()})
        np.savez_compressed(os.path.join(OUT, "step_%s.npz" % tag), **rec)


def gen_laploss():
    """kwatsch/lap_pyramid_loss.py (imports torch only): pyramid levels, loss and input gradient of the reference's LapLoss on
    1-channel (the trainers' use, base_trainer.py:54) and 2-channel inputs."""
    import kwatsch.lap_pyramid_loss as lp
    rec = {}
    for tag, shape in (("a", (3, 1, 32, 32)), ("b", (2, 2, 24, 24))):      # square only: upsample() :27-34 mixes H and W
        g = torch.Generator().manual_seed(sum(shape))
        x = torch.rand(shape, generator=g).requires_grad_(True)
        t = torch.rand(shape, generator=g)
        crit = lp.LapLoss(max_levels=3, channels=shape[1], device="cpu")
        pyr = lp.laplacian_pyramid(x, crit.gauss_kernel, 3)
        loss = crit(x, t)
        loss.backward()
        rec.update({tag + "/x": x.detach().numpy(), tag + "/t": t.numpy(), tag + "/loss": np.float64(loss.item()),
                    tag + "/dx": x.grad.numpy()})
        for k, p in enumerate(pyr):
            rec["%s/pyr%d" % (tag, k)] = p.detach().numpy()
    np.savez_compressed(os.path.join(OUT, "laploss.npz"), **rec)


def gen_ae_standard_blocks():
    """networks/ae_standard.py:34-80: one BasicEncoderBlock (conv, LReLU, conv, LReLU, AvgPool2d; no BatchNorm) followed by
    one BasicDecoderBlock (conv, LReLU, conv, LReLU, bilinear Upsample x2): forward, input gradient, parameter gradients.
    The module imports without shims; its AE class needs a missing `networks.model_configs`, the blocks do not."""
    import networks.ae_standard as ast
    for tag, (N, H, W, cin, cmid) in {"a": (2, 12, 16, 8, 16), "b": (3, 10, 6, 4, 8)}.items():
        torch.manual_seed(4242 + N)
        enc = ast.BasicEncoderBlock(cin, cmid, kernel=3, padding=1, downsample=True, use_batchnorm=False)
        dec = ast.BasicDecoderBlock(cmid, cin, kernel=3, padding=1, do_upsample=True)
        for m in list(enc.modules()) + list(dec.modules()):
            ast.weights_init(m)
            if isinstance(m, nn.Conv2d):
                m.bias.data.normal_(std=0.1)
        x = torch.rand(N, cin, H, W, requires_grad=True)
        tgt = torch.randn(N, cin, H, W)
        mid = enc(x)
        out = dec(mid)
        loss = (out * tgt).mean() + 0.5 * (out ** 2).mean()
        loss.backward()
        rec = {"x": x.detach().numpy().copy(), "tgt": tgt.numpy().copy(), "mid": mid.detach().numpy().copy(),
               "out": out.detach().numpy().copy(), "loss": np.float64(loss.item()), "dx": x.grad.numpy().copy()}
        for name, mod in (("enc", enc), ("dec", dec)):
            for k, p_ in mod.named_parameters():
                if k.startswith("batchnorm"):
                    continue
                rec["p/%s.%s" % (name, k)] = p_.detach().numpy().copy()
                rec["grad/%s.%s" % (name, k)] = p_.grad.numpy().copy()
        np.savez_compressed(os.path.join(OUT, "ae_standard_blocks_%s.npz" % tag), **rec)


def gen_augmentation():
    """The reference's ACDC training transforms (train_cardiac_aesr.py:83-96) on synthetic triplets: inputs, the seed of the
    numpy RandomState the transforms draw from, and the outputs.  datasets/shared_transforms.py needs cv2 / batchgenerators /
    torchvision at import time only (elastic transforms this path never builds): stubbed."""
    for name in ("cv2", "batchgenerators", "batchgenerators.transforms"):
        if name not in sys.modules:
            _stub(name)
    _stub("batchgenerators.transforms.spatial_transforms", SpatialTransform=object)
    _stub("batchgenerators.transforms.abstract_transforms", Compose=object)
    if "torchvision" not in sys.modules:
        tv = _stub("torchvision")
        tv.transforms = _stub("torchvision.transforms")
    import datasets.shared_transforms as st
    rec = {}
    cases = {"a": (3, (54, 64), 40, 32, 11), "b": (4, (35, 38), 40, 32, 12), "c": (2, (40, 40), 40, 40, 13),
             "d": (3, (25, 58), 46, 16, 14)}
    for tag, (n, (H, W), aug, width, seed) in cases.items():
        g = np.random.RandomState(1000 + seed)
        rs = np.random.RandomState(seed)
        chain = [st.AdjustToPatchSize((aug, aug)), st.CenterCrop((aug, aug)), st.RandomCrop(width, rs=rs), st.RandomIntensity(rs=rs),
                 st.RandomRotation(rs)]
        ins, outs = [], []
        for _ in range(n):
            trip = g.rand(3, H, W).astype(np.float32)
            sample = {"image": trip.copy()}
            for t in chain:
                sample = t(sample)
            ins.append(trip)
            outs.append(np.ascontiguousarray(sample["image"]).astype(np.float32))
        rec["%s/in" % tag] = np.stack(ins)
        rec["%s/out" % tag] = np.stack(outs)
        rec["%s/cfg" % tag] = np.array([aug, width, seed], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "augment_acdc.npz"), **rec)


def gen_eval_crop():
    """AdjustToPatchSize + CenterCrop as the evaluation uses them (evaluate/find_best_model.py:26-33) on small volumes."""
    for name in ("cv2", "batchgenerators", "batchgenerators.transforms"):
        if name not in sys.modules:
            _stub(name)
    _stub("batchgenerators.transforms.spatial_transforms", SpatialTransform=object)
    _stub("batchgenerators.transforms.abstract_transforms", Compose=object)
    import datasets.shared_transforms as st
    rec = {}
    for i, (shape, ps) in enumerate((((2, 20, 28), 32), ((2, 40, 52), 32), ((2, 33, 31), 32), ((1, 64, 64), 48), ((2, 29, 40), 32))):
        v = np.random.RandomState(50 + i).rand(*shape).astype(np.float32)
        out = st.CenterCrop(ps)(st.AdjustToPatchSize((ps, ps))({"image": v.copy()}))["image"]
        rec["%d/in" % i], rec["%d/out" % i], rec["%d/ps" % i] = v, np.ascontiguousarray(out), np.array(ps)
    np.savez_compressed(os.path.join(OUT, "eval_crop.npz"), **rec)


def gen_vif():
    """VIF as the reference's evaluation computes it: ``evaluate.metrics.compute_vif_for_batch`` (:65-109: both volumes -> uint8, then
    ``evaluate.vifvec.vifp_mscale`` per slice) and ``vifp_mscale`` itself (:7-63) on uint8 / float32 / float64 slices, run as the
    reference's OWN functions.  ``evaluate/metrics.py`` imports skimage / datasets / lpips at module level (never called here):
    absent modules are answered with empty stand-ins."""
    import_reference()
    tv = sys.modules["torchvision"]
    for sub in ("datasets", "transforms", "utils"):
        setattr(tv, sub, _any_stub("torchvision." + sub))
    em = vv = None
    for _ in range(64):
        try:
            import evaluate.metrics as em
            import evaluate.vifvec as vv
            break
        except ModuleNotFoundError as e:
            _any_stub(e.name)
    rs = np.random.RandomState(4711)

    def smooth_volume(z, h, w, noise):
        """MRI-like slices in [0, 1]: blobs on a black background (exact zeros), a saturated patch (exact ones), and a degraded copy."""
        yy, xx = np.mgrid[0:h, 0:w]
        vol = np.zeros((z, h, w), dtype=np.float64)
        for k in range(z):
            for _ in range(6):
                cy, cx, sg, amp = rs.uniform(0.2 * h, 0.8 * h), rs.uniform(0.2 * w, 0.8 * w), rs.uniform(0.05, 0.25) * min(h, w), rs.uniform(0.2, 0.9)
                vol[k] += amp * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * sg * sg))
            vol[k] += 0.05 * rs.randn(h, w)
            vol[k][(yy - h / 2) ** 2 + (xx - w / 2) ** 2 > (0.45 * min(h, w)) ** 2] = 0.0
        vol = np.clip(vol, 0, 1)
        vol[:, h // 3:h // 3 + 6, w // 3:w // 3 + 9] = 1.0
        deg = np.clip(0.9 * vol + noise * rs.randn(z, h, w) + 0.02, 0, 1)
        deg[vol == 0] = 0.0
        return vol.astype(np.float32), deg.astype(np.float32)

    rec = {}
    cases = [("v40", 5, 40, 52, 0.04, None), ("v33", 4, 33, 31, 0.08, None), ("v64ds", 9, 64, 64, 0.03, 2), ("v28", 3, 28, 28, 0.1, None),
             ("v72ds3", 7, 72, 56, 0.02, 3)]
    for tag, z, h, w, noise, ds in cases:
        a, b = smooth_volume(z, h, w, noise)
        rec[tag + "/ref"], rec[tag + "/dist"] = a, b
        rec[tag + "/downsample_steps"] = np.array(-1 if ds is None else ds)
        rec[tag + "/vif_batch"] = np.array(em.compute_vif_for_batch(a, b, eval_axis=0, normalize=False, downsample_steps=ds), dtype=np.float64)
        a8, b8 = np.uint8(np.clip(a * 255., 0, 255)), np.uint8(np.clip(b * 255., 0, 255))
        rec[tag + "/vif_u8"] = np.array([vv.vifp_mscale(a8[k], b8[k]) for k in range(z)], dtype=np.float64)
        rec[tag + "/vif_f32"] = np.array([vv.vifp_mscale(a[k], b[k]) for k in range(z)], dtype=np.float64)
        rec[tag + "/vif_f64"] = np.array([vv.vifp_mscale(a[k].astype(np.float64), b[k].astype(np.float64)) for k in range(z)], dtype=np.float64)
    # a single 2-D image pair (the reference returns the score itself) and the degenerate ones: identical images, a black image
    a, b = smooth_volume(1, 48, 48, 0.05)
    rec["img/ref"], rec["img/dist"] = a[0], b[0]
    rec["img/vif_batch"] = np.array(em.compute_vif_for_batch(a[0], b[0]), dtype=np.float64)
    rec["same/vif_batch"] = np.array(em.compute_vif_for_batch(a[0], a[0]), dtype=np.float64)
    with np.errstate(all="ignore"):
        rec["black/vif_batch"] = np.array(em.compute_vif_for_batch(np.zeros((2, 24, 24), np.float32), np.zeros((2, 24, 24), np.float32)), dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "vif.npz"), **rec)
    print("vif.npz:", {k: (v.tolist() if v.size < 10 else v.shape) for k, v in rec.items() if "vif" in k})


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(4)
    if len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "vif":
        gen_vif()
        return
    if len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "eval_crop":
        import_reference()
        gen_eval_crop()
        return
    if len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "augment":
        import_reference()
        gen_augmentation()
        return
    if len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "ae_standard":
        import_reference()
        gen_ae_standard_blocks()
        return
    if len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "acai":
        gen_acai_steps(import_reference()[0])
        return
    if len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "laploss":
        import_reference()
        gen_laploss()
        return
    if len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "trainer_steps":
        gen_trainer_steps(import_reference()[0])
        return
    if len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "val_volumes":
        gen_val_volumes(import_reference()[0])
        return
    if len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "supervolume_eval":
        gen_supervolume_eval(import_reference()[0])
        return
    av, avs, avm, nb = import_reference()
    gen_supervolume_eval(av)
    gen_vif()
    gen_acai_steps(av)
    gen_laploss()
    gen_ae_standard_blocks()
    gen_augmentation()
    gen_eval_crop()
    gen_ae_small(av, avs, avm)
    m = gen_ae_init(av)
    gen_ae_acdc_probe(m)
    gen_lpips(nb)
    gen_trainer_steps(av)          # step_k3_*: the reference's own trainer classes (supersedes the restated gen_steps)
    gen_supervolume(av)
    gen_val_volumes(av)            # last: it rebinds names inside the reference's evaluate.* modules
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("wrote %d fixtures, %.1f KiB" % (len(os.listdir(OUT)), tot / 1024))


if __name__ == "__main__":
    main()


# tests/golden/net_config.json (plugin table of networks/net_config.py for every (model, dataset, ae_class)) is produced by:
#   python - <<'PY'
#   import sys, json; sys.path.insert(0, '/root/reference'); import networks.net_config as nc
#   rows = {}
#   for net in ['ae','aesr','ae_combined','aesr_combined','vae','vae_combined','vae2','acai','acai_combined']:
#       for ds in [None,'ACDC','ACDCLBL','dHCP','ADNI','OASIS','MNIST3D','MNISTRoto']:
#           for ae in ['VanillaACAI','LargerAE','VanillaACAIStrided']:
#               try: rows['%s|%s|%s' % (net, ds, ae)] = nc.NetworkConfig(net, dataset=ds, ae_class=ae).architecture
#               except ValueError: rows['%s|%s|%s' % (net, ds, ae)] = 'ValueError'
#   json.dump({'MODULE_PATH': nc.MODULE_PATH, 'rows': rows}, open('tests/golden/net_config.json', 'w'), indent=0, sort_keys=True)
#   PY
