"""``get_trainer_dynamic``: importlib-based trainer factory with the reference's calling convention
(kwatsch/get_trainer.py:23-85).  Train: ``get_trainer_dynamic(args_dict, model_file=...) -> trainer``.
Eval: ``get_trainer_dynamic(src_path=..., model_nbr=..., eval_mode=True) -> (trainer, args_dict)`` reading
``<src_path>/settings.yaml`` and ``<src_path>/models/<nbr>.models``.  The module paths / class names come from
``networks.net_config`` (or from settings.yaml of an existing experiment) and are resolved unchanged."""
import os
from importlib import import_module

from .common import load_settings


def _module_name(path):
    name = path.replace("/", ".").replace(".py", "")
    return name.replace("utils.", "kwatsch.")        # backward compatibility of old settings files (reference :73)


def _prepare(args_dict, src_path, model_nbr, model_file, kwargs):
    model_file_sr = None
    if src_path is not None:
        src_path = os.path.expanduser(src_path)
        args_dict = load_settings(os.path.join(src_path, "settings.yaml"))
        args_dict.setdefault("output_dir", src_path)
        model_file = os.path.expanduser(os.path.join(src_path, "models", "{}.models".format(model_nbr)))
        if kwargs.get("model_nbr_sr") is not None:
            model_file_sr = os.path.expanduser(os.path.join(src_path, "models", "{}.models".format(kwargs["model_nbr_sr"])))
    args_dict.setdefault("use_extra_latent_loss", False)
    args_dict.setdefault("use_alpha_probe", False)
    args_dict.setdefault("alpha_dims", None)
    return args_dict, model_file, model_file_sr


def get_trainer_dynamic(args_dict=None, src_path=None, model_nbr=None, eval_mode=False, args_only=False, model_file=None,
                        **kwargs):
    if model_nbr is None and args_dict is None:
        raise ValueError("ERROR - get_trainer - args_dict or model_filename needs to be specified")
    if model_file is not None:
        print("Warning - get trainer - RETRAIN model {}".format(model_file))
    args_dict, model_file, model_file_sr = _prepare(args_dict, src_path, model_nbr, model_file, kwargs)
    if "device" in kwargs:                              # additive: run an old experiment on another device string
        args_dict["device"] = kwargs["device"]
    if args_only:
        return None, args_dict
    ae_class_name = args_dict.get("ae_class", "VanillaACAI").replace("default", "VanillaACAI")
    ae_class = getattr(import_module(_module_name(args_dict["module_network_path"])), ae_class_name)
    ae_model = ae_class(args_dict).to(args_dict["device"])
    aesr_model = None if model_file_sr is None else ae_class(args_dict).to(args_dict["device"])
    trainer_mod = import_module(_module_name(args_dict["module_trainer_path"]))
    trainer_cls = getattr(trainer_mod, args_dict.get("trainer_class", "AEBaseTrainer"))
    trainer = trainer_cls(args_dict, ae_model, model_file=model_file, eval_mode=eval_mode, model_sr=aesr_model,
                          model_file_sr=model_file_sr)
    return trainer if src_path is None else (trainer, args_dict)


def get_trainer(args_dict=None, src_path=None, model_nbr=None, eval_mode=False, args_only=False, model_file=None):
    """Legacy selector by ``args_dict['model']`` (reference :88-181) for the models of this build."""
    if src_path is not None:
        _, args_dict = get_trainer_dynamic(src_path=src_path, model_nbr=model_nbr, args_only=True)
    model = (args_dict or {}).get("model", "").lower()
    if model not in ("ae", "ae_combined", "acai", "acai_combined"):
        raise ValueError("Error - get trainer - no trainer available for model {}".format(model))
    from ..networks.net_config import NetworkConfig
    cfg = NetworkConfig(model, dataset=args_dict.get("dataset", "ACDC"), ae_class="VanillaACAI").architecture
    for k in ("module_network_path", "module_trainer_path", "trainer_class"):
        args_dict.setdefault(k, cfg[k])
    return get_trainer_dynamic(args_dict=None if src_path else args_dict, src_path=src_path, model_nbr=model_nbr,
                               eval_mode=eval_mode, args_only=args_only, model_file=model_file)
