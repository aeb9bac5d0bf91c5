"""Import-path shim: re-exports superresolution_aniso_mri_amd.evaluate.evaluate_image (reference module evaluate/evaluate_image.py)."""
import importlib as _il

_impl = _il.import_module("superresolution_aniso_mri_amd.evaluate.evaluate_image")
globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
