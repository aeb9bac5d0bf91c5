"""Import-path shim (settings.yaml stores `networks/acai_vanilla_strided.py`): re-exports superresolution_aniso_mri_amd.networks.acai_vanilla_strided."""
import importlib as _il

_impl = _il.import_module("superresolution_aniso_mri_amd.networks.acai_vanilla_strided")
globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
