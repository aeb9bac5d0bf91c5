#!/usr/bin/env python
"""Drop-in entry point (same flags as the reference's generate_hr_volumes.py:186-209), served by the MI355X build."""
from superresolution_aniso_mri_amd.generate_hr_volumes import *  # noqa: F401,F403
from superresolution_aniso_mri_amd.generate_hr_volumes import main

if __name__ == "__main__":
    main()
