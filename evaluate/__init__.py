"""Import-path shim: the reference's top-level package name, served by superresolution_aniso_mri_amd."""
