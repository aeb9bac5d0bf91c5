"""Import-path shim (settings.yaml stores `kwatsch/base_trainer.py`): re-exports superresolution_aniso_mri_amd.kwatsch.base_trainer."""
import importlib as _il

_impl = _il.import_module("superresolution_aniso_mri_amd.kwatsch.base_trainer")
globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
