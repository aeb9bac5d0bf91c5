"""Brain trainers (reference kwatsch/brain/trainer_ae.py:8-281): per-sample mixing coefficients
``alpha_from`` / ``alpha_to`` ([B,1], from the slice distances) instead of the cardiac 0.5 / 0.5."""
import os

import torch

from .. import trainer_ae as _tae
from ..cardiac.trainer_ae import CombinedStepMixin


class AEBaseTrainerBrain(_tae.AEBaseTrainer):
    _per_sample_alpha = True

    def _mix_coefficients(self, batch_item, B):
        if batch_item is None or "alpha_from" not in batch_item:
            return 0.5, 0.5
        dev = self.args["device"]
        return (batch_item["alpha_from"].to(dev).float().reshape(-1), batch_item["alpha_to"].to(dev).float().reshape(-1))

    def _pred_alphas(self, batch_item):
        return batch_item["alpha_from"].detach() if "alpha_from" in batch_item else torch.tensor([0.5])


class AETrainerBrain(AEBaseTrainerBrain):
    """plain ``ae`` on brain data: AEBaseTrainer.train with per-sample alphas (reference :50-89)."""


class AETrainerExtension1Brain(CombinedStepMixin, AEBaseTrainerBrain):
    """``ae_combined`` on dHCP / OASIS / ADNI (reference :92-281)."""
    _mask_inputs = False
    _log_extra_total = False

    def _extra_weight(self):
        return self.args["ex_loss_weight1"]          # no annealing in the brain trainer (reference :163-165)

    def _pred_alphas(self, batch_item):
        return torch.tensor([0.5])

    def validate(self, validation_batch, image_dict=None, frame_id=8, generate_images=True):
        res = super().validate(validation_batch, image_dict=image_dict, frame_id=frame_id, generate_images=generate_images)
        self._validate_synthesis(validation_batch, add_to_loss_ae=True)
        if self.epoch > self.args["epoch_threshold"]:
            self.save_best_val_model()
        return res

    def save_best_val_model(self, **kwargs):
        super().save_best_val_model()
        if self._best_now("loss_ae_dist_extra"):
            self.save_models(os.path.join(self.args["dir_models"], "caisr.models"), self.epoch + 1)
