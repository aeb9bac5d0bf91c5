#!/usr/bin/env python
"""Drop-in entry point with the flags of the reference's train_cardiac_aesr.py (e.g.
``python train_cardiac_aesr.py --dataset=ACDC --model=ae_combined --batch_size=12 --latent=128 --latent_width=32 --width=128
--downsample_steps=2 --epochs=900 --ex_loss_weight1=0.05 --aug_patch_size=160 --exper_id=x --output_dir=/tmp/expers --synthetic
--synthetic_size=160``)."""
from superresolution_aniso_mri_amd.train_aesr import main, merge_args_architecture  # noqa: F401

if __name__ == "__main__":
    main()
