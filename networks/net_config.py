"""Import-path shim (settings.yaml stores `networks/net_config.py`): re-exports superresolution_aniso_mri_amd.networks.net_config."""
import importlib as _il

_impl = _il.import_module("superresolution_aniso_mri_amd.networks.net_config")
globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
