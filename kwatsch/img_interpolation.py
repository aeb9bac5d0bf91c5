"""Import-path shim (settings.yaml stores `kwatsch/img_interpolation.py`): re-exports superresolution_aniso_mri_amd.kwatsch.img_interpolation."""
import importlib as _il

_impl = _il.import_module("superresolution_aniso_mri_amd.kwatsch.img_interpolation")
globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
