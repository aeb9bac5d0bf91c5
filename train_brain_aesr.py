#!/usr/bin/env python
"""Drop-in entry point with the flags of the reference's train_brain_aesr.py (dHCP / OASIS / ADNI / MNISTRoto)."""
from superresolution_aniso_mri_amd.train_aesr import main, merge_args_architecture  # noqa: F401

if __name__ == "__main__":
    main(brain=True)
