"""MNIST trainers (reference kwatsch/mnist/trainer_ae.py:4-71)."""
from ..brain.trainer_ae import AEBaseTrainerBrain, AETrainerExtension1Brain


class AECombinedTrainerMNIST(AETrainerExtension1Brain):
    """``ae_combined`` on MNISTRoto / MNIST3D: the brain step with optional loss annealing (reference :48-71)."""

    def _extra_weight(self):
        if self.args.get("use_loss_annealing"):
            return float(self.loss_weights[self.epoch])
        return self.args["ex_loss_weight1"]


class AETrainerMNIST(AEBaseTrainerBrain):
    """plain ``ae`` on MNIST (reference :6-45): only the reconstruction loss is optimised; the synthesised slice and
    the latent loss are produced for logging, with the per-sample mixing coefficients."""
