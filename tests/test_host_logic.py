"""not-gpu: host-side logic of the drop-in boundary -- C-ABI export table, plugin table, CLI flags, trainer factory,
checkpoint / settings formats, loss logging, annealing table, batch sharding.  No kernel is launched here."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def test_cabi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "aesr_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(aesr_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    from superresolution_aniso_mri_amd import _hip
    lib = ctypes.CDLL(_hip.LIB_PATH)
    for name in declared:
        assert getattr(lib, name, None) is not None, "%s is declared in include/aesr_hip.h but not exported" % name
    assert declared == set(_hip.SIGNATURES.keys()), declared ^ set(_hip.SIGNATURES.keys())
    assert _hip.lib.aesr_version() == 1
    # argument validation happens on the host before any launch -> callable without a GPU
    assert _hip.lib.aesr_lerp_fwd(None, None, None, None, 1, 4, None) != 0
    assert "aesr_lerp_fwd" in _hip.last_error()
    assert _hip.lib.aesr_conv2d_packed_floats(32, 32, 3, 0) == 9 * 32 * 32
    assert _hip.lib.aesr_conv2d_packed_floats(1, 32, 3, 0) == 9 * 32 * 16          # Cout padded to one MFMA block
    assert _hip.lib.aesr_conv2d_wgrad_workspace_floats(24, 160, 160, 32, 32, 3, 1) > 0
    # which Winograd kernel serves a layer (host-side rule, conv_wino_res.hip): resident filter for K-side channels <= 32, and
    # <= 64 where 8 x 8-output blocks tile the image with <= 10 % padding; the ring kernel (conv_wino_ring.hip; AESR_WINO_RING=0: the
    # first streamed kernel) otherwise; 0 = not a Winograd layer
    k = _hip.lib.aesr_conv2d_wino_kernel
    if os.environ.get("AESR_WINO_RES") is None:
        assert k(36, 160, 160, 32, 32, 3, 1, 0) == 2 and k(36, 81, 81, 32, 64, 3, 1, 0) == 2
        assert k(36, 80, 80, 64, 32, 3, 1, 0) == 2
        assert k(36, 80, 80, 32, 64, 3, 1, 1) == 2          # data gradient: the K side is Cout
        if os.environ.get("AESR_WINO_RING") is None:
            # streamed layers: all on the ring kernel (the default) ...
            assert k(36, 40, 40, 128, 128, 3, 1, 0) == 3 and k(24, 80, 80, 128, 128, 3, 1, 0) == 3 and k(24, 20, 20, 512, 512, 3, 1, 0) == 3
            assert k(36, 81, 81, 64, 64, 3, 1, 0) == 3
            # ... with AESR_WINO_RING=1 only where its cost estimate is lower: the first streamed kernel on 81 x 81 (18 % block padding);
            # VGG conv5 (less than a round of ring items) goes to the ring kernel with its 32 chunks split over 4 items per block group,
            # for which the library asks for 4 output-sized slabs
            os.environ["AESR_WINO_RING"] = "1"
            try:
                assert k(36, 81, 81, 64, 64, 3, 1, 0) == 1 and k(36, 40, 40, 128, 128, 3, 1, 0) == 3
            finally:
                del os.environ["AESR_WINO_RING"]
            assert k(24, 10, 10, 512, 512, 3, 1, 0) == 3
            wsf = _hip.lib.aesr_conv2d_wino_workspace_floats
            assert wsf(24, 10, 10, 512, 512, 0) == 4 * 24 * 10 * 10 * 512 and wsf(36, 40, 40, 128, 128, 0) == 0
            assert wsf(2, 10, 10, 512, 512, 1) == 16 * 2 * 10 * 10 * 512          # a 1-triplet shard: 16 splits of 2 chunks
    assert k(36, 160, 160, 3, 64, 3, 1, 0) == 0 and k(36, 160, 160, 32, 32, 1, 0, 0) == 0


def test_net_config_matches_reference_table():
    from networks import net_config as nc          # through the import-path shim settings.yaml relies on
    g = json.load(open(os.path.join(GOLDEN, "net_config.json")))
    assert g["MODULE_PATH"] == nc.MODULE_PATH
    for key, want in g["rows"].items():
        net, ds, ae = key.split("|")
        ds = None if ds == "None" else ds
        try:
            got = nc.NetworkConfig(net, dataset=ds, ae_class=ae).architecture
        except ValueError:
            got = "ValueError"
        assert got == want, key


def test_arguments_defaults_and_derived():
    from kwatsch.arguments import parse_args
    args, d = parse_args(["--dataset=ACDC", "--model=ae_combined", "--batch_size=12", "--latent=128", "--latent_width=32",
                          "--width=128", "--downsample_steps=2", "--ex_loss_weight1=0.05", "--aug_patch_size=160",
                          "--exper_id=x", "--output_dir=/tmp/o"])
    assert d["image_mix_loss_func"] == "perceptual" and d["lr"] == 1e-5 and d["seed"] == 892372 and d["epochs"] == 250
    assert d["output_dir"] == "/tmp/o/x" and d["vae_beta"] == 0 and d["lamb"] == 0 and d["test_batch_size"] == 16
    _, d = parse_args(["--dataset=OASIS", "--model=ae_combined", "--width=64", "--downsample_steps=4"])
    assert d["aug_patch_size"] == 220
    _, d = parse_args(["--model=ae", "--downsample_steps=2"])
    assert d["aug_patch_size"] == 180 and d["image_mix_loss_func"] is None
    with pytest.raises(ValueError):
        parse_args(["--model=ae"])


def _cpu_args(tmp, **kw):
    from networks.net_config import NetworkConfig
    args = dict(model="ae_combined", dataset="ACDC", device="cpu", lr=1e-5, weight_decay=0.0, epochs=20, width=32, latent_width=8,
                depth=8, latent=16, ex_loss_weight1=0.05, use_percept_loss=False, get_masks=False, use_loss_annealing=False,
                use_extra_latent_loss=False, epoch_threshold=0, ae_class="VanillaACAI", image_mix_loss_func="mse",
                output_dir=str(tmp), dir_models=str(tmp), dir_images=str(tmp))
    args.update(kw)
    for k, v in NetworkConfig("ae_combined", dataset=args["dataset"], ae_class=args["ae_class"]).architecture.items():
        args.setdefault(k, v)
    return args


def test_get_trainer_dynamic_resolves_reference_paths(tmp_path):
    from kwatsch.common import saveExperimentSettings
    from kwatsch.get_trainer import get_trainer_dynamic
    for ds, cls in (("ACDC", "AETrainerEndToEnd"), ("OASIS", "AETrainerExtension1Brain"), ("MNISTRoto", "AECombinedTrainerMNIST")):
        tr = get_trainer_dynamic(_cpu_args(tmp_path, dataset=ds), eval_mode=True)
        assert type(tr).__name__ == cls and type(tr.model).__name__ == "VanillaACAI"
        assert tr.iters == 1 and tr.epoch == 0 and tr.percept_criterion is None and tr.eval_fixed_coeff
    tr = get_trainer_dynamic(_cpu_args(tmp_path, ae_class="LargerAE"), eval_mode=True)
    assert type(tr.model).__name__ == "LargerAE"
    with pytest.raises(ValueError):
        get_trainer_dynamic()
    # eval-mode convention: (trainer, args_dict) from <src>/settings.yaml + <src>/models/<nbr>.models
    tr = get_trainer_dynamic(_cpu_args(tmp_path), eval_mode=True)
    os.makedirs(tmp_path / "models", exist_ok=True)
    tr.args["dir_models"] = str(tmp_path / "models")
    tr.save_models(str(tmp_path / "models" / "7.models"), 7)
    saveExperimentSettings(tr.args, str(tmp_path / "settings.yaml"))
    tr2, a2 = get_trainer_dynamic(src_path=str(tmp_path), model_nbr=7, eval_mode=True)
    assert a2["module_trainer_path"] == "kwatsch/cardiac/trainer_ae.py" and tr2.model_file.endswith("models/7.models")
    for (k, a), (_, b) in zip(tr.model.state_dict().items(), tr2.model.state_dict().items()):
        assert torch.equal(a, b), k
    ck = torch.load(str(tmp_path / "models" / "7.models"))
    assert set(ck) == {"model_dict_ae", "optimizer_dict_ae", "epoch"}
    want = ["enc.%d.%s" % (i, n) for i in (0, 1, 3) for n in ("weight", "bias")] + ["enc.5.weight", "enc.5.bias",
            "enc.5.running_mean", "enc.5.running_var", "enc.5.num_batches_tracked"]
    assert list(ck["model_dict_ae"].keys())[:11] == want           # SURVEY App. B (nn.Sequential indices)
    _, a3 = get_trainer_dynamic(src_path=str(tmp_path), model_nbr=7, args_only=True)
    assert a3["width"] == 32


def test_product_init_is_the_reference_init():
    from networks.acai_vanilla import VanillaACAI
    rec = dict(np.load(os.path.join(GOLDEN, "ae_init_acdc.npz")))
    torch.manual_seed(892372)
    m = VanillaACAI(dict(width=128, latent_width=32, depth=32, latent=128, colors=1, use_batchnorm=True, use_sigmoid=True,
                         device="cpu"))
    assert sum(p.numel() for p in m.parameters()) == 443777
    for k, p in m.named_parameters():
        assert np.array_equal(p.detach().flatten()[:4].numpy(), rec["head/" + k]), k


def test_no_cpu_fallback_anywhere(tmp_path):
    from kwatsch.get_trainer import get_trainer_dynamic
    from superresolution_aniso_mri_amd.data_synth import synthetic_batch
    tr = get_trainer_dynamic(_cpu_args(tmp_path), eval_mode=True)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        tr.train(synthetic_batch(2, 32, 32, seed=0))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        tr.encode(torch.rand(1, 1, 32, 32))


def test_loss_log_and_annealing(tmp_path):
    from kwatsch.base_trainer import LossLog
    from kwatsch.get_trainer import get_trainer_dynamic
    log = LossLog()
    log.append(torch.tensor(1.5))
    log.append(2.5)
    assert log[-1] == 2.5 and log[0] == 1.5 and log.floats() == [1.5, 2.5] and isinstance(list.__getitem__(log, 0), float)
    tr = get_trainer_dynamic(_cpu_args(tmp_path, epochs=20, use_loss_annealing=True), eval_mode=True)
    x = np.linspace(-5, 5, 20)
    want = (torch.sigmoid(torch.from_numpy(x)) * 0.05).numpy()[::-1]       # kwatsch/base_trainer.py:456-459
    np.testing.assert_allclose(tr.loss_weights, want, rtol=1e-12)
    assert tr._extra_weight() == pytest.approx(float(want[0]))
    tr._log("loss_ae", torch.tensor(0.25))
    tr._log("loss_ae", 0.75)
    tr.show_loss_on_tensorboard()
    assert tr.mean_losses["loss_ae"] == [0.5] and tr.loss_iters == [1]
    tr.reset_losses()
    assert len(tr.losses["loss_ae"]) == 0
    tr.save_losses()
    it, ltr, lte = tr.load_losses(str(tmp_path))
    assert list(it) == [1] and float(ltr["loss_ae"][0]) == 0.5


def test_synthetic_batch_layout_and_sharding():
    from superresolution_aniso_mri_amd.data_synth import shard_batch, synthetic_batch
    b = synthetic_batch(12, 40, 40, seed=3, brain=True)
    assert b["image"].shape == (24, 1, 40, 40) and b["slice_between"].shape == (12, 1, 40, 40) and b["alpha_from"].shape == (12, 1)
    assert torch.equal(synthetic_batch(12, 40, 40, seed=3, brain=True)["image"], b["image"])
    assert float(b["image"].min()) >= 0 and float(b["image"].max()) <= 1
    sizes = []
    for r in range(8):
        s = shard_batch(b, r, 8)
        n = s["slice_between"].shape[0]
        sizes.append(n)
        assert s["image"].shape[0] == 2 * n
        lo = (12 * r) // 8
        assert torch.equal(s["image"][:n], b["image"][lo:lo + n]) and torch.equal(s["image"][n:], b["image"][12 + lo:12 + lo + n])
        assert torch.equal(s["alpha_to"], b["alpha_to"][lo:lo + n])
    assert sorted(sizes) == [1, 1, 1, 1, 2, 2, 2, 2] and sum(sizes) == 12


def test_make_grid_and_recon_grid():
    from kwatsch.acai_utils import generate_recon_grid, make_grid
    t = torch.arange(6 * 1 * 4 * 5, dtype=torch.float32).reshape(6, 1, 4, 5)
    g = make_grid(t, 3, padding=2, pad_value=0.5)
    assert g.shape == (1, 2 * 6 + 2, 3 * 7 + 2) and float(g[0, 0, 0]) == 0.5 and float(g[0, 2, 2]) == 0.0
    assert torch.equal(g[0, 8:12, 2:7], t[3, 0])
    assert generate_recon_grid(torch.rand(4, 1, 8, 8), torch.rand(4, 1, 8, 8)).shape[0] == 1


def test_original_slice_ids_for_subsampled_volumes():
    """evaluate/metrics.py:29-45: which slices of a sub-sampled-then-upsampled volume are originals (hand-derived cases)."""
    import importlib.util, os, sys, types
    # the module imports the HIP binding at import time only for the device functions; load just the pure function
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "superresolution_aniso_mri_amd", "evaluate", "metrics.py")).read()
    start = src.index("def determine_original_sliceids"); end = src.index("def _as_volume")
    ns = {"np": np}
    exec(src[start:end], ns)
    f = ns["determine_original_sliceids"]
    assert list(f(np.zeros((13, 4, 4)), 2)) == [0, 2, 4, 6, 8, 10, 12]
    assert list(f(np.zeros((10, 4, 4)), 3)) == [0, 3, 6, 9]
    assert list(f(np.zeros((11, 4, 4)), 3)) == [0, 3, 6, 9, 10]
    assert list(f(np.zeros((11, 4, 4)), 3, conv_interpol=True)) == [0, 3, 6, 9, 10]
    assert list(f(np.zeros((12, 4, 4)), 4, conv_interpol=True)) == [0, 4, 8, 9, 10, 11]


def test_evaluate_common_host_helpers(tmp_path):
    """evaluate/common.py:11-38: result file naming, last kept slice, stripping of conventionally expanded volumes."""
    from evaluate import common as ec
    rec = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "supervolume_eval.npz"))
    for n, d, v in rec["determine_last_slice"]:
        assert ec.determine_last_slice(n, d) == v
    sr, orig = np.arange(10)[:, None, None] * np.ones((1, 2, 2)), 100 + np.arange(4)[:, None, None] * np.ones((1, 2, 2))
    out = ec.strip_conventional_interpolation_results(sr, orig, 3)
    assert out.shape == (8, 2, 2) and out[-1, 0, 0] == 103 and out[-2, 0, 0] == 6
    ec.save_metrics(str(tmp_path), "ACDC", {"ssim": np.array([0.5])}, 2, "ae_combined", 0)
    assert os.path.isfile(str(tmp_path / "results" / "ACDC_ae_combined_2x.npz"))
    ec.save_metrics(str(tmp_path), None, {"ssim": np.array([0.5])}, 3, "linear", 1)
    assert os.path.isfile(str(tmp_path / "results" / "linear_3x_axis1.npz"))
    with pytest.raises(NotImplementedError):
        ec.create_simple_interpolation(np.zeros((2, 2, 2)), np.ones(3), expand_factor=2)


def test_evaluation_crop_and_masks_vs_reference_transforms():
    """evaluate/find_best_model.adjust_and_center_crop == CenterCrop(AdjustToPatchSize(.)) of the reference's transform classes
    (tests/golden/eval_crop.npz, incl. the padding rule that uses patch_size[0] for both axes); synthesis / reconstruction masks."""
    from evaluate.find_best_model import adjust_and_center_crop, generate_synth_slices_mask, get_transforms, store_top_scores
    rec = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "eval_crop.npz"))
    for i in range(5):
        got = adjust_and_center_crop(rec["%d/in" % i], int(rec["%d/ps" % i]))
        assert got.shape == rec["%d/out" % i].shape and np.array_equal(got, rec["%d/out" % i]), i
    t = get_transforms(32, to_tensor=True)({"image": rec["0/in"], "patient_id": 7})
    assert torch.is_tensor(t["image"]) and t["patient_id"] == 7 and tuple(t["image"].shape) == (2, 32, 32)
    r_mask, s_mask = generate_synth_slices_mask(9, 3)           # 9 slices, every 3rd kept: last paired slice = 6
    assert r_mask.tolist() == [True, False, False, True, False, False, True] and (r_mask ^ s_mask).all()
    top = store_top_scores("3", {}, [0.5, 0.7], [20.0, 30.0], [float("nan")] * 2)
    assert top["3"][0] == pytest.approx(0.6) and top["3"][1] == 25.0 and np.isnan(top["3"][2])


def test_largest_activation_of_a_pass_and_planner_queries():
    """Host logic of the inference path: the bound that decides how many images one decoder pass takes is the largest tensor the compiled
    steps really make (folded upsampling, pooling, stem), and the eval-mode epilogue query follows the planner without touching a device."""
    from superresolution_aniso_mri_amd import _hip, engine
    from superresolution_aniso_mri_amd.networks import acai_vanilla as av
    dec = engine.SequentialRunner(av.Decoder(2, 32, 128, 1, use_batchnorm=True))
    assert [s.kind for s in dec.steps] == ["conv", "conv", "bn", "conv", "conv", "bn", "conv", "conv"]
    assert dec.max_elems_per_image(56, 56, 128) == 224 * 224 * 32                  # the 32-channel tensors at full resolution
    assert dec.max_elems_per_image(56, 56, 64, first=1) == 224 * 224 * 32         # ... also behind the decoder's first convolution
    assert dec.max_elems_per_image(56, 56, 128, last=2) == 56 * 56 * 128           # the input itself where nothing bigger follows
    enc = engine.SequentialRunner(av.Encoder(2, 32, 128, 1, use_batchnorm=True))
    assert enc.max_elems_per_image(224, 224, 1) == 226 * 226 * 32                  # the 1x1 / padding-1 stem grows the image by 2
    assert (1 << 28) // dec.max_elems_per_image(56, 56, 64, first=1) >= 29 * 3     # a dHCP volume (30 slices, 3 mixes per pair): one pass
    L = _hip.lib
    assert L.aesr_conv2d_wino_fwd_bn_supported(30, 226, 226, 32, 32) == 1          # resident-filter layer
    assert L.aesr_conv2d_wino_fwd_bn_supported(4, 16, 16, 24, 32) == 0             # not a Winograd layer
    assert L.aesr_conv2d_wino_fwd_bn_supported(0, 16, 16, 32, 32) == 0
    for shape in [(30, 113, 113, 64, 64), (2, 40, 40, 128, 128), (1, 20, 20, 256, 256)]:
        kind = L.aesr_conv2d_wino_kernel(*shape, 3, 1, 0)
        if L.aesr_conv2d_wino_fwd_bn_supported(*shape):
            assert kind in (2, 3)
        else:
            assert kind in (1, 3)        # the first streamed kernel, or the ring kernel only WITH a channel split
