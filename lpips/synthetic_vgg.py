"""Import-path shim: re-exports superresolution_aniso_mri_amd.lpips.synthetic_vgg."""
import importlib as _il

_impl = _il.import_module("superresolution_aniso_mri_amd.lpips.synthetic_vgg")
globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
