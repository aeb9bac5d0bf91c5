"""Plugin table of the reference (networks/net_config.py:2-7, :19-92): which module file / trainer class serves a
(model, dataset, ae_class) combination plus the architecture defaults that ``train_*_aesr.py`` merges under the CLI
arguments.  The strings are persisted in ``settings.yaml`` and resolved by ``kwatsch.get_trainer`` with importlib, so
the paths name the same module files as the reference; rows whose trainers are outside the ae/ae_combined hot path
keep their reference strings (they resolve only if such a module is supplied by the user)."""

MODULE_PATH = {
    "VanillaACAI": "networks/acai_vanilla.py",
    "VAE": "networks/beta_vae.py",
    "VAE2": "networks/beta_vae.py",
    "LargerAE": "networks/acai_vanilla_modified.py",
    "MultiChannelAE": "networks/acai_multi_channel.py",
    "VanillaACAIStrided": "networks/acai_vanilla_strided.py",
}

BRAIN_SETS = ("dHCP", "ADNI", "OASIS")
MNIST_SETS = ("MNIST3D", "MNISTRoto")

# (network family, dataset key) -> (module_trainer_path, trainer_class, extra architecture entries)
_TRAINERS = {
    ("ae", "ACDC"): ("kwatsch/trainer_ae.py", "AEBaseTrainer", {}),
    ("ae", None): ("kwatsch/trainer_ae.py", "AEBaseTrainer", {}),
    ("ae", "ACDCLBL"): ("kwatsch/sr_multi_channel/trainer_ae.py", "MultiChannelTrainer", {"nclasses": 4}),
    ("ae", "brain"): ("kwatsch/brain/trainer_ae.py", "AETrainerBrain", {}),
    ("ae", "mnist"): ("kwatsch/mnist/trainer_ae.py", "AETrainerMNIST", {}),
    ("ae_combined", "ACDC"): ("kwatsch/cardiac/trainer_ae.py", "AETrainerEndToEnd", {}),
    ("ae_combined", "ACDCLBL"): ("kwatsch/sr_multi_channel/trainer_ae.py", "MultiChannelCAISRTrainer", {"nclasses": 4}),
    ("ae_combined", "brain"): ("kwatsch/brain/trainer_ae.py", "AETrainerExtension1Brain", {}),
    ("ae_combined", "mnist"): ("kwatsch/mnist/trainer_ae.py", "AECombinedTrainerMNIST", {}),
}


def _dataset_key(dataset):
    if dataset in BRAIN_SETS:
        return "brain"
    if dataset in MNIST_SETS:
        return "mnist"
    return dataset


class NetworkConfig(object):

    def __init__(self, network, dataset=None, ae_class="VanillaACAI"):
        self.network = network
        self.dataset = dataset
        self.ae_class = ae_class
        self.architecture = {}
        self.load_config()

    def load_config(self):
        arch = self.architecture
        arch.update(width=128, latent_width=16, depth=32, colors=2 if self.dataset == "ACDCLBL" else 1, latent=16,
                    use_laploss=False, use_percept_loss=False, n_res_block=None, use_batchnorm=True, use_sigmoid=True,
                    max_grad_norm=0, fine_tune=False, ex_loss_weight1=0.5,
                    module_network_path=MODULE_PATH[self.ae_class])
        net = self.network
        unsupported = ValueError("Error - NetworkConfig - Unsupported combination {}/{}".format(net, self.dataset))
        if net in ("ae", "aesr", "ae_combined", "aesr_combined"):
            family = "ae_combined" if "combined" in net else "ae"
            key = (family, _dataset_key(self.dataset))
            if key not in _TRAINERS:
                raise unsupported
            path, cls, extra = _TRAINERS[key]
            arch.update(module_trainer_path=path, trainer_class=cls, **extra)
            # plain AE: no synthesis loss; combined: LPIPS on the synthesised slice (reference :51,:53)
            arch["image_mix_loss_func"] = "perceptual" if family == "ae_combined" else None
        elif net in ("vae", "vae_combined", "vae2", "acai", "acai_combined"):
            if self.dataset not in MNIST_SETS + ("ACDC",) + BRAIN_SETS:
                raise ValueError("Error - network {} does not support dataset {}".format(net.upper(), self.dataset))
            arch["image_mix_loss_func"] = "perceptual" if "combined" in net else None
            if net.startswith("vae"):
                arch.update(module_trainer_path="kwatsch/trainer_vae.py", trainer_class="VAETrainer")
            else:
                arch.update(module_trainer_path="kwatsch/trainer_acai.py", trainer_class="ACAITrainer")
