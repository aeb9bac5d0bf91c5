"""Import-path shim: re-exports superresolution_aniso_mri_amd.evaluate.find_best_model (reference module evaluate/find_best_model.py)."""
import importlib as _il

_impl = _il.import_module("superresolution_aniso_mri_amd.evaluate.find_best_model")
globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
