"""Import-path shim (settings.yaml stores `kwatsch/mnist/trainer_ae.py`): re-exports superresolution_aniso_mri_amd.kwatsch.mnist.trainer_ae."""
import importlib as _il

_impl = _il.import_module("superresolution_aniso_mri_amd.kwatsch.mnist.trainer_ae")
globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
