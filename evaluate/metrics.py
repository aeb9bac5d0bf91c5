"""Import-path shim: re-exports superresolution_aniso_mri_amd.evaluate.metrics (reference module evaluate/metrics.py)."""
import importlib as _il

_impl = _il.import_module("superresolution_aniso_mri_amd.evaluate.metrics")
globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
