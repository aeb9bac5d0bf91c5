"""Import-path shim: re-exports superresolution_aniso_mri_amd.kwatsch.lap_pyramid_loss (reference module kwatsch/lap_pyramid_loss.py)."""
import importlib as _il

_impl = _il.import_module("superresolution_aniso_mri_amd.kwatsch.lap_pyramid_loss")
globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
