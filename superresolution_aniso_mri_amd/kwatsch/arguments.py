"""Command-line flags of the training scripts, same names / defaults / derived values as the reference's
kwatsch/arguments.py:10-103, plus additive flags for the MI355X build (data parallel, synthetic data, step graph)."""
import argparse
import os

DATASETS = ["ACDC", "ACDCC", "dHCP", "ADNI", "OASIS", "MNIST3D", "MNISTRoto", "ACDCLBL"]
MODELS = ["ae", "ae_combined", "aesr", "aesr_combined", "vae", "vae_combined", "acai", "acai_combined", "vae2"]


def build_parser():
    p = argparse.ArgumentParser(description="Train auto-encoder for SR")
    a = p.add_argument
    a("-d", "--dataset", type=str, choices=DATASETS, default="ACDC", help="Dataset to train on")
    a("--downsample_steps", type=int, default=None)      # through-plane anisotropy factor of the data
    a("-ss", "--slice_selection", type=str, choices=["adjacent_plus", "adjacent", "mix"], default="adjacent_plus")
    a("-c", "--comment", type=str, default=None)
    a("-m", "--model", type=str, choices=MODELS, default="ae", help="Model to train")
    a("-id", "--exper_id", type=str, default="debug", help="Determine subdir where output is stored")
    a("-o", "--output_dir", type=str, default=None)
    a("--model_filename", type=str, default=None)
    a("-e", "--epochs", type=int, default=250)
    a("-l", "--lr", type=float, default=0.00001)
    a("-w", "--weight_decay", type=float, default=0.)
    a("-b", "--batch_size", type=int, default=12)
    a("-bt", "--test_batch_size", type=int, default=16)
    a("--device", type=str, default="cuda")
    a("--limited_load", action="store_true")
    a("-s", "--seed", type=int, default=892372)
    a("-g", "--gpu_ids", type=int, nargs="+", default=[0])
    a("-p", "--port", type=int, default=8030)
    a("--number_of_workers", type=int, default=2)
    a("--validate_every", type=int, default=500)
    a("--alpha_loss_func", type=str, default=None, choices=[None, "mse", "perceptual"])
    for flag in ("use_percept_loss", "use_ssim_loss", "use_extra_latent_loss", "use_loss_annealing"):
        a("--" + flag, action="store_true")
    a("--alpha_class", type=str, default=None)
    a("--width", type=int, default=128)
    a("--latent_width", type=int, default=16)
    a("--latent", type=int, default=16)
    a("--depth", type=int, default=32)
    a("--ae_class", type=str, default="VanillaACAI")
    a("--image_mix_loss_func", type=str, default=None)
    a("--ex_loss_weight1", type=float, default=0.001)
    a("--lamb_reg_acai", type=float, default=0.5)
    a("--vae_beta", type=float, default=None)
    a("--aug_patch_size", type=int, default=None)
    a("--get_masks", action="store_true")
    a("--log_tensorboard", action="store_true")
    a("--epoch_threshold", type=int, default=100, help="save models > epoch_threshold")
    # ---- additive (MI355X build) ----
    a("--synthetic", action="store_true", help="train on synthetic triplets (no dataset on disk needed)")
    a("--synthetic_size", type=int, default=None, help="H=W of the synthetic slices (default: --width)")
    a("--synthetic_pool", type=int, default=8, help="distinct device-resident synthetic batches that --synthetic cycles through (made before the loop)")
    a("--volumes_dir", type=str, default=None, help="train on the volumes (.npy [Z,H,W] / [T,Z,H,W], .nii(.gz), .mha, .mhd) of this "
                                                  "directory: device-resident cache + on-device triplet assembly / augmentation")
    a("--val_volumes_dir", type=str, default=None, help="4-D (or 3-D) validation images of this directory become the in-memory image_dict "
                                                      "that validate() previews as val_image_e###_p###.png (at most --val_patients of them)")
    a("--val_patients", type=int, default=2, help="validation patients kept for the whole-volume previews (the reference loads 2)")
    a("--iters_per_epoch", type=int, default=50, help="iterations per epoch with --synthetic")
    a("--vgg_weights", type=str, default=None, help="local torchvision vgg16 state_dict for LPIPS (offline)")
    a("--use_step_graph", action="store_true", help="capture the training step in a HIP graph")
    return p


def finalize_args(args):
    """Derived defaults (reference kwatsch/arguments.py:64-103)."""
    if args.model == "ae_combined" and args.image_mix_loss_func is None:
        args.image_mix_loss_func = "perceptual"
        print("!!! Warning !!! - arguments - Using perceptual loss for image mix distance")
    if args.model in ("vae", "vae_combined"):
        args.ae_class = "VAE"
        args.lamb = 1.
        if args.model == "vae" and args.vae_beta is None:
            args.vae_beta = 100
    elif args.model == "vae2":
        args.ae_class, args.lamb = "VAE2", 1
        if args.vae_beta is None:
            args.vae_beta = 1
    else:
        args.vae_beta, args.lamb = 0, 0
    if args.downsample_steps is None:
        raise ValueError("Error - arguments - downsample_steps cannot be None")
    forced = {"OASIS": 220, "dHCP": 256}
    if args.dataset in forced and args.aug_patch_size is None and args.width < forced[args.dataset]:
        args.aug_patch_size = forced[args.dataset]
    if args.dataset in ("ACDC", "ACDCLBL") and args.aug_patch_size is None:
        args.aug_patch_size = 180
    if args.output_dir is not None:
        args.output_dir = os.path.expanduser(os.path.join(args.output_dir, args.exper_id))
    else:
        args.output_dir = os.path.expanduser(os.path.join("~/expers/sr_redo", args.dataset, args.model, args.exper_id))
    if args.model_filename is not None:
        args.model_filename = os.path.expanduser(args.model_filename)
    return args


def parse_args(argv=None):
    args = finalize_args(build_parser().parse_args(argv))
    return args, vars(args)
