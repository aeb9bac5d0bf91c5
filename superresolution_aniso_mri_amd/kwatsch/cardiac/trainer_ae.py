"""``AETrainerEndToEnd``: the ACDC ``ae_combined`` step of the reference (kwatsch/cardiac/trainer_ae.py:8-182).

One step = loss_ae(MSE(dec(enc(x)), x)) + lambda * loss(slice_between, dec(0.5 z_from + 0.5 z_to)), with the
reference's four network passes (enc(x[2B]), dec(z[2B]), dec(z_mix[B]), logging-only enc(slice_between[B]) which still
updates the encoder's BatchNorm running statistics) issued as TWO batched launches sequences with
per-sub-batch BatchNorm statistics: enc([x | slice_between]) and dec([z | z_mix])."""
import os

import torch

from .. import trainer_ae as _tae
from ... import ops


class CombinedStepMixin(object):
    """Shared ``ae_combined`` machinery of the cardiac / brain / MNIST trainers."""

    def _extra_weight(self):
        if self.args.get("use_loss_annealing"):
            return float(self.loss_weights[self.epoch])
        return self.args["ex_loss_weight1"]

    def get_extra_image_loss(self, reference, synthesized, mask=None, is_test=False):
        use_mask = self.args.get("get_masks") and mask is not None
        if self.percept_criterion is not None and self.image_mix_loss_func == "perceptual":
            if use_mask and self._mask_inputs:
                m = mask[:synthesized.size(0)].to(synthesized.device)
                reference, synthesized = reference * m, synthesized * m
            ctxm = torch.no_grad() if is_test else torch.enable_grad()
            with ctxm:
                d = self.percept_criterion(reference, synthesized, normalize=True)
                if use_mask and not self._mask_inputs:
                    d = d * mask[:d.size(0)].to(d.device)
                return d.mean()
        if use_mask:
            # reference :117-120: mean over ALL elements of mse(reduction='none') * mask (rarely used option: plain tensor ops)
            m = mask[:synthesized.size(0)].to(synthesized.device)
            loss_image = ((reference - synthesized) ** 2 * m).mean()
        else:
            loss_image = ops.mse_loss(reference, synthesized)
        if getattr(self, "laploss", None) is not None:                      # reference :124-125 (MSE branch only)
            loss_image = self.laploss(synthesized, reference) + loss_image
        return loss_image

    _mask_inputs = True      # cardiac multiplies the images by the mask; brain multiplies the distances

    _log_extra_total = True

    def synthesize_batch_images(self, **kwargs):
        """Eval-style entry kept for callers (validate): lerp + decode (+ latent loss against enc(slice_between))."""
        batch_item, z, between = kwargs.get("batch_item"), kwargs.get("z"), kwargs.get("slice_between", None)
        is_eval = kwargs.get("is_eval", False)
        z = z.to(self.args["device"])
        a_from, a_to = self._mix_coefficients(batch_item, z.shape[0] // 2)
        z_mix = ops.lerp_mix(z, a_from, a_to)
        if is_eval:
            s_mix = self.decode(z_mix, eval=True)
            self.model.train()
        else:
            s_mix = self.model.decode(z_mix)
        lat = 0
        if kwargs.get("compute_latent_loss", False):
            z_ref = self.encode(between, eval=is_eval)
            lat = ops.mse_loss(z_mix.detach(), z_ref.detach())
        return {"s_between_mix": s_mix, "z_mix": z_mix, "loss_latent": lat}

    def _lambda_tensor(self):
        """Synthesis-loss weight as a device scalar (so an annealed weight can change under a captured step graph)."""
        w = float(self._extra_weight())
        t = self.__dict__.get("_lam_t")
        if t is None:
            t = self._lam_t = torch.empty((), dtype=torch.float32, device=self.args["device"])
            self._lam_host = None
        if self._lam_host != w:
            t.fill_(w)
            self._lam_host = w
        return t

    def get_extra_loss(self, slice_between, s_between_mix, z_mix, z=None, mask=None, is_test=False):
        loss_img = self._lambda_tensor() * self.get_extra_image_loss(slice_between, s_between_mix, mask=mask, is_test=is_test)
        if self.args.get("use_extra_latent_loss"):
            raise NotImplementedError("use_extra_latent_loss is not part of the README ae_combined recipes")
        if self._log_extra_total:
            self._log("loss_ae_extra", loss_img, is_test)
        self._log("loss_ae_dist_extra", loss_img, is_test)
        return loss_img

    def _fused_mse_losses(self):
        """True when every loss of the step is a plain mean-squared error (README recipe ae_combined with the MSE synthesis
        loss: no LPIPS, no Laplacian pyramid, no masks): the loss block then runs as ops.combined_mse.  AESR_FUSED_LOSS=0 keeps
        the term-by-term path."""
        if os.environ.get("AESR_FUSED_LOSS", "1") == "0":
            return False
        percept = self.percept_criterion is not None and ("perceptual" in (self.ae_loss_func, self.image_mix_loss_func))
        return not (percept or getattr(self, "laploss", None) is not None or self.args.get("get_masks")
                    or self.args.get("use_extra_latent_loss"))

    def _step_core(self, batch_item, eval_mode):
        """The ae_combined step on device-resident inputs; logs through ``_log`` and returns the tensors callers keep."""
        x, between = batch_item["image"], batch_item["slice_between"]
        B = x.shape[0] // 2
        if hasattr(self.model, "prepare_weights"):
            self.model.prepare_weights()          # encoder + decoder operands after the last optimizer step: ONE launch
        if self.dp is not None and self.dp.active:
            self.dp.begin_step()                  # the peer exchange of the SyncBN sums (AESR_SYNCBN=p2p) advances its generation
        # enc(x[2B]) and the logging-only enc(slice_between[B]): one batched pass, two BatchNorm statistic groups
        z, z_ref = self.model.encode_multi([x, between], needs_grad=[True, False])
        a_from, a_to = self._mix_coefficients(batch_item, B)
        # dec(z[2B]) and dec(z_mix[B]): one batched pass, two statistic groups; its input [z | z_mix] comes from one kernel
        zcat = ops.lerp_cat(z, a_from, a_to)
        z_mix = zcat[2 * B:]
        if self._fused_mse_losses():
            # all three mean-squared errors, their weighted sum and (backward) the whole gradient of the decoder output: two
            # launches instead of fifteen small ones (ops.combined_mse); same quantities, same log keys, same order
            o3 = self.model.decode_cat(zcat, [2 * B, B], merge=True)[0]
            out, s_mix = o3[:2 * B], o3[2 * B:]
            loss, l_rec, l_img, loss_latent = ops.combined_mse(o3, x, between, z_mix.detach(), z_ref.detach(), self._lambda_tensor(), owner=self)
            self._log("loss_ae_dist", l_rec)
            if self._log_extra_total:
                self._log("loss_ae_extra", l_img)
            self._log("loss_ae_dist_extra", l_img)
            self._backward_and_step(loss, eval_mode)
            self._log("loss_ae", loss)
            self._log("loss_latent_1", loss_latent)
            return {"z_mix": z_mix, "s_mix": s_mix, "out": out}
        out, s_mix = self.model.decode_cat(zcat, [2 * B, B])
        loss_ae = self.get_loss(x, out, is_test=False)["loss_ae"]
        loss_latent = ops.mse_loss(z_mix.detach(), z_ref.detach())
        mask = batch_item["loss_mask"] if self.args.get("get_masks") else None
        loss = loss_ae + self.get_extra_loss(between, s_mix, z_mix, z=z, mask=mask, is_test=False)
        self._backward_and_step(loss, eval_mode)
        self._log("loss_ae", loss)
        self._log("loss_latent_1", loss_latent)
        return {"z_mix": z_mix, "s_mix": s_mix, "out": out}

    def train(self, batch_item, keep_predictions=True, eval_mode=False):
        dev_batch = {k: (self._to_device(v) if k in ("image", "slice_between", "alpha_from", "alpha_to") else v)
                     for k, v in batch_item.items()}
        self._set_mode(not eval_mode)
        self._iters += 1
        self._note_batch(batch_item, "train")
        self._lambda_tensor()
        if self._graph_ok(keep_predictions, eval_mode):
            self._train_graphed(dev_batch)
            return
        r = self._step_core(dev_batch, eval_mode)
        if keep_predictions:
            self._bounded_sync()          # data parallel: the host copies below wait for the step; a dead peer must not hang them
            s = r["s_mix"].detach().cpu()
            self.train_predictions = {"z_mix": r["z_mix"].detach().cpu(), "pred_alphas": self._pred_alphas(batch_item),
                                      "slice_inbetween_mix": s, "slice_inbetween_05": s, "reconstruction": r["out"].detach().cpu()}

    def _pred_alphas(self, batch_item):
        return torch.tensor([0.5])

    def _validate_synthesis(self, validation_batch, add_to_loss_ae):
        z = self.test_predictions["z"].to(self.args["device"])
        between = self._to_device(validation_batch["slice_between"])
        r = self.synthesize_batch_images(batch_item=validation_batch, z=z, compute_latent_loss=True, slice_between=between,
                                         is_eval=True)
        self._log("loss_latent_1", r["loss_latent"], True)
        mask = validation_batch["loss_mask"] if self.args.get("get_masks") else None
        extra = self.get_extra_loss(between, r["s_between_mix"], r["z_mix"], mask=mask, is_test=True)
        if add_to_loss_ae:
            self.losses_test["loss_ae"][-1] = self.losses_test["loss_ae"][-1] + float(extra)


class AETrainerEndToEnd(CombinedStepMixin, _tae.AEBaseTrainer):

    def validate(self, validation_batch, image_dict=None, frame_id=8, generate_images=True):
        res = super().validate(validation_batch, image_dict=image_dict, frame_id=frame_id, generate_images=generate_images)
        self._validate_synthesis(validation_batch, add_to_loss_ae=False)
        return res

    def save_best_val_model(self, **kwargs):
        if self._best_now("loss_ae_dist_extra"):
            self.save_models(os.path.join(self.args["dir_models"], "caisr.models"), self.epoch + 1)
