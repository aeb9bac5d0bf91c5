"""Import-path shim (settings.yaml stores `kwatsch/trainer_acai.py`): re-exports superresolution_aniso_mri_amd.kwatsch.trainer_acai."""
import importlib as _il

_impl = _il.import_module("superresolution_aniso_mri_amd.kwatsch.trainer_acai")
globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
