"""-m gpu: the drop-in scripts end to end on synthetic data: train_cardiac_aesr.py writes settings.yaml + .models,
generate_hr_volumes.py reloads them through get_trainer_dynamic(src_path=..., model_nbr=...) and super-resolves a volume.
Also BASELINE config 1 (MNIST-shaped 28x28, latent 16, B=32, LPIPS) as one trainer step against the CPU oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_train_then_generate(tmp_path):
    from superresolution_aniso_mri_amd import generate_hr_volumes, train_aesr
    out = str(tmp_path / "expers")
    tr = train_aesr.main(["--dataset=ACDC", "--model=ae_combined", "--batch_size=4", "--test_batch_size=4", "--latent=16",
                          "--latent_width=8", "--width=32", "--depth=8", "--downsample_steps=2", "--epochs=2", "--lr=0.001",
                          "--ex_loss_weight1=0.05", "--exper_id=t1", "--output_dir=" + out, "--synthetic", "--iters_per_epoch=3",
                          "--image_mix_loss_func=mse", "--epoch_threshold=0"])
    src = os.path.join(out, "t1")
    assert os.path.isfile(os.path.join(src, "settings.yaml")) and os.path.isfile(os.path.join(src, "models", "2.models"))
    assert tr.iters == 1 + 6 and len(tr.mean_losses["loss_ae"]) >= 1
    assert os.path.isfile(os.path.join(src, "losses_train.npz"))
    data = tmp_path / "vols"
    data.mkdir()
    np.save(str(data / "vol0.npy"), np.random.RandomState(0).rand(5, 32, 32).astype(np.float32) * 900.0)   # needs percentile normalisation
    res = generate_hr_volumes.main(["--exper_dir=" + src, "--model_nbr=2", "--num_interpolations=3", "--data_input_dir=" + str(data),
                                    "--output_dir=" + str(tmp_path / "hr"), "--save"])
    hr = np.load(str(tmp_path / "hr" / "vol0.npy"))
    assert hr.shape == (4 * 4 + 1, 32, 32) and hr.min() >= 0 and hr.max() <= 1 and len(res) == 1
    # the same volume as a NIfTI file (no SimpleITK in this image: volume_io.py reads / writes it) gives the same slices, a
    # float32 .nii.gz with 17 slices and the z spacing divided by 4
    import struct
    from superresolution_aniso_mri_amd import volume_io
    vol = np.random.RandomState(0).rand(5, 32, 32).astype(np.float32) * 900.0
    nii = tmp_path / "nii"
    nii.mkdir()
    like = volume_io.Volume(vol, (1.5, 1.5, 8.0), "npy", {})
    volume_io.write_volume(nii / "vol0.nii.gz", like, vol, (1.5, 1.5, 8.0))
    generate_hr_volumes.main(["--exper_dir=" + src, "--model_nbr=2", "--num_interpolations=3", "--data_input_dir=" + str(nii),
                              "--output_dir=" + str(tmp_path / "hr_nii"), "--save"])
    back = volume_io.read_volume(tmp_path / "hr_nii" / "vol0.nii.gz")
    assert back.array.shape == (17, 32, 32) and back.array.dtype == np.float32
    assert abs(back.spacing[2] - 2.0) < 1e-6 and back.spacing[:2] == (1.5, 1.5)
    assert np.array_equal(back.array, hr)


def test_train_from_volume_directory(tmp_path):
    """--volumes_dir: NIfTI + npy volumes -> device-resident cache -> on-device triplet assembly / augmentation -> training."""
    from superresolution_aniso_mri_amd import train_aesr, volume_io
    data = tmp_path / "vols"
    data.mkdir()
    g = np.random.RandomState(3)
    v = (g.rand(9, 44, 52) * 1200).astype(np.float32)
    volume_io.write_volume(data / "p1.nii.gz", volume_io.Volume(v, (1.4, 1.4, 8.0), "npy", {}), v, (1.4, 1.4, 8.0))
    np.save(str(data / "p2.npy"), g.rand(2, 8, 40, 40).astype(np.float32))          # 4-D: two frames
    out = str(tmp_path / "expers")
    tr = train_aesr.main(["--dataset=ACDC", "--model=ae_combined", "--batch_size=4", "--test_batch_size=4", "--latent=16",
                          "--latent_width=8", "--width=32", "--depth=8", "--downsample_steps=2", "--epochs=2", "--lr=0.001",
                          "--ex_loss_weight1=0.05", "--exper_id=v1", "--output_dir=" + out, "--volumes_dir=" + str(data),
                          "--val_volumes_dir=" + str(data),
                          "--aug_patch_size=40", "--iters_per_epoch=3", "--image_mix_loss_func=mse", "--epoch_threshold=0",
                          "--use_step_graph"])
    assert tr.iters == 1 + 6 and np.isfinite(tr.mean_losses["loss_ae"][-1])
    assert os.path.isfile(os.path.join(out, "v1", "models", "2.models"))
    # --val_volumes_dir: the two images of the directory are the in-memory validation patients; every epoch leaves one preview each
    import glob
    pngs = sorted(os.path.basename(f) for f in glob.glob(os.path.join(out, "v1", "**", "val_image_e*_p*.png"), recursive=True))
    assert pngs == ["val_image_e%03d_p%03d.png" % (e, p) for e in (1, 2) for p in (1, 2)], pngs


def test_config1_mnist_shaped_step_vs_oracle():
    from oracle import ae_oracle, lpips_oracle, step_oracle
    from superresolution_aniso_mri_amd.data_synth import synthetic_batch
    from superresolution_aniso_mri_amd.kwatsch.get_trainer import get_trainer_dynamic
    from superresolution_aniso_mri_amd.networks.net_config import NetworkConfig
    cfg = dict(width=28, latent_width=7, depth=32, latent=16, colors=1, use_batchnorm=True, use_sigmoid=True)
    args = dict(model="ae_combined", dataset="MNISTRoto", device="cuda", lr=1e-5, weight_decay=0.0, epochs=2, ex_loss_weight1=0.001,
                use_percept_loss=False, get_masks=False, use_loss_annealing=False, use_extra_latent_loss=False, epoch_threshold=100,
                ae_class="VanillaACAI", image_mix_loss_func="perceptual", vgg_weights="synthetic-hash", **cfg)
    for k, v in NetworkConfig("ae_combined", dataset="MNISTRoto").architecture.items():
        args.setdefault(k, v)
    torch.manual_seed(1)
    trainer = get_trainer_dynamic(args)
    assert type(trainer).__name__ == "AECombinedTrainerMNIST"
    oracle = ae_oracle.OracleAE(cfg, init=False).load_state_dict({k: v.detach().cpu() for k, v in trainer.model.state_dict().items()})
    lin = np.load(os.path.join(os.path.dirname(__file__), "..", "superresolution_aniso_mri_amd", "lpips", "weights", "v0.1", "vgg_lin.npz"))
    ost = step_oracle.OracleStep(oracle, lr=1e-5, ex_loss_weight1=0.001, image_mix_loss_func="perceptual",
                                 vgg_sd=lpips_oracle.hash_vgg16_state(),
                                 lin_w=[torch.from_numpy(lin["lin%d" % k]).reshape(1, -1, 1, 1) for k in range(5)])
    batch = synthetic_batch(32, 28, 28, seed=9, brain=True)
    trainer.train(batch)
    ref = ost.train(batch["image"], batch["slice_between"], batch["alpha_from"], batch["alpha_to"])
    for key, want in (("loss_ae", ref["loss_ae"]), ("loss_ae_dist", ref["loss_ae_dist"]), ("loss_ae_dist_extra", ref["loss_ae_dist_extra"]),
                      ("loss_latent_1", ref["loss_latent_1"])):
        assert abs(trainer.losses[key][-1] - want) <= 3e-5 * abs(want), key
    rel = float((trainer.train_predictions["reconstruction"].double() - ref["out"].double()).norm() / ref["out"].double().norm())
    assert rel < 1e-5


@pytest.mark.parametrize("syncbn", ["rccl", "p2p"])
def test_two_rank_training_cli_on_one_device(tmp_path, syncbn):
    """train_cardiac_aesr.py under torch.distributed.run with TWO ranks rehearsed on the one device of the test box (gloo data plane; with
    AESR_SYNCBN=p2p the SyncBN sums travel through IPC-mapped peer regions inside the one-launch BatchNorm kernels): the loop, its deadline
    syncs, the cross-rank epoch means and the rank-0 checkpoints; both exchange forms print the same epoch losses."""
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AESR_SINGLE_DEVICE="1", AESR_DIST_BACKEND="gloo", AESR_SYNCBN=syncbn, AESR_BN_FUSED_NB="64", AESR_P2P_SPINS=str(1 << 18))
    out = str(tmp_path / "expers")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "train_cardiac_aesr.py"), "--dataset=ACDC", "--model=ae_combined", "--batch_size=4", "--test_batch_size=4", "--latent=16",
           "--latent_width=8", "--width=32", "--depth=8", "--downsample_steps=2", "--epochs=2", "--lr=0.001", "--ex_loss_weight1=0.05", "--exper_id=dp",
           "--output_dir=" + out, "--synthetic", "--iters_per_epoch=4", "--image_mix_loss_func=mse", "--epoch_threshold=0", "--use_step_graph"]
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("epoch")]
    assert len(lines) == 2 and all("nan" not in ln for ln in lines), r.stdout[-1500:]
    assert os.path.isfile(os.path.join(out, "dp", "models", "2.models")) and os.path.isfile(os.path.join(out, "dp", "settings.yaml"))
    # both exchange forms compute the same global batch statistics: the printed epoch means agree to the printed digits
    ref = tmp_path.parent / "two_rank_cli_lines.txt"
    if ref.exists():
        assert ref.read_text() == "\n".join(lines)
    else:
        ref.write_text("\n".join(lines))


def test_bench_launches_its_own_ranks():
    """``python bench.py --gpus 2`` with NO outer launcher (WORLD_SIZE unset): the ranks start as fresh child processes under
    torch.distributed.run before the parent touches torch or the GPU; rank 0's single JSON line and the exit code come through.  Two
    ranks rehearsed on the one device of the test box (gloo data plane)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(AESR_SINGLE_DEVICE="1", AESR_DIST_BACKEND="gloo", AESR_BN_FUSED_NB="64")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--no-secondary"],
                       env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-1500:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["value"] > 0 and line["scaling"] == "strong"
    assert "graph form segments" in line["config"]["launch"] and "SyncBN exchange" in line["config"]["launch"]
    # a rank that fails takes the launcher's exit code with it and no line is printed
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-secondary",
                          "--config", "nope"], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and bad.stdout.strip() == ""


def test_two_rank_training_from_a_volume_directory_equals_single_process(tmp_path):
    """train_cardiac_aesr.py --volumes_dir under torch.distributed.run with two ranks rehearsed on one device: every rank draws the whole global
    batch's random numbers and assembles only its own triplets (TripletAugmenter.next_batch(shard=...)) into the buffer its captured step reads;
    with SyncBN and the weighted gradient all-reduce the epoch means equal the single-process run's."""
    import re
    import socket
    import subprocess
    import sys
    from superresolution_aniso_mri_amd import volume_io
    data = tmp_path / "vols"
    data.mkdir()
    g = np.random.RandomState(11)
    for i in range(3):
        np.save(str(data / ("v%d.npy" % i)), g.rand(9, 44, 48).astype(np.float32))
    v = (g.rand(8, 40, 52) * 900).astype(np.float32)
    volume_io.write_volume(data / "p7.nii.gz", volume_io.Volume(v, (1.4, 1.4, 8.0), "npy", {}), v, (1.4, 1.4, 8.0))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = ["--dataset=ACDC", "--model=ae_combined", "--batch_size=5", "--test_batch_size=4", "--latent=16", "--latent_width=8", "--width=32",
            "--depth=8", "--downsample_steps=2", "--epochs=2", "--lr=0.001", "--ex_loss_weight1=0.05", "--volumes_dir=" + str(data),
            "--aug_patch_size=40", "--iters_per_epoch=4", "--image_mix_loss_func=mse", "--epoch_threshold=0", "--use_step_graph"]

    def epoch_lines(r):
        assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("epoch")]
        assert len(lines) == 2 and all("nan" not in ln for ln in lines), r.stdout[-1500:]
        return [[float(x) for x in re.findall(r"[-+]?\d+\.\d+(?:e[-+]?\d+)?", ln)] for ln in lines]

    env1 = {k: v_ for k, v_ in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    one = subprocess.run([sys.executable, os.path.join(root, "train_cardiac_aesr.py")] + args + ["--exper_id=one", "--output_dir=" + str(tmp_path / "e1")],
                         env=env1, cwd=root, capture_output=True, text=True, timeout=400)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env2 = dict(env1, AESR_SINGLE_DEVICE="1", AESR_DIST_BACKEND="gloo")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(root, "train_cardiac_aesr.py")] + args +
                         ["--exper_id=two", "--output_dir=" + str(tmp_path / "e2")], env=env2, cwd=root, capture_output=True, text=True, timeout=400)
    a, b = epoch_lines(one), epoch_lines(two)
    for la, lb in zip(a, b):
        assert len(la) == len(lb) and len(la) >= 2
        np.testing.assert_allclose(lb, la, rtol=2e-3, atol=1e-6)        # lr 1e-3: Adam's sign noise separates the trajectories at the 1e-3 level
