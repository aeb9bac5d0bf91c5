"""``ACAITrainer``: the adversarially-constrained auto-encoder step of the reference (kwatsch/trainer_acai.py:34-160) on the HIP
engine -- models ``acai`` and ``acai_combined`` of networks/net_config.py.

Per step (reference :46-96, :98-127):
    z = enc(x[2B]);  out = dec(z);                       loss_ae_dist = MSE(x, out)
    d_reg = critic(out + gamma*(x - out));               loss_disc_l2 = mean(d_reg^2)            gamma = 0.2
    alpha ~ U[0, 0.5)^B;  out_mix = dec(alpha*z[:B] + (1-alpha)*z[B:]);  d_mix = critic(out_mix)
    loss_ae_l2 = mean(d_mix^2);                          loss_disc_dist = MSE(d_mix, alpha)
    loss_ae   = loss_ae_dist + lamb_reg_acai * loss_ae_l2 (+ the ae_combined synthesis loss for ``acai_combined``)
    loss_disc = loss_disc_dist + loss_disc_l2
The reference calls ``loss_ae.backward(retain_graph=True)`` then ``loss_disc.backward()`` without detaching anything, so every
parameter of BOTH networks receives d(loss_ae + loss_disc): here that is one backward pass of the sum.  The three decoder
passes (z, acai mix, synthesis mix) and the two critic passes run as single launch sequences with per-sub-batch BatchNorm
statistics, in the reference's update order.  alpha is drawn from the host generator exactly where the reference draws it, so
the step cannot be replayed from a captured graph."""
import torch

from .. import ops
from .base_trainer import optimizer_state_to_cpu
from .brain.trainer_ae import AETrainerExtension1Brain


def swap_halves(x):
    a, b = x.split(x.shape[0] // 2)
    return torch.cat([b, a])


def lerp(start, end, weights):
    return start + weights * (end - start)


def interp_image(image1, image2, alpha):
    return alpha * image1 + ((1 - alpha) * image2)


class ACAITrainer(AETrainerExtension1Brain):

    def __init__(self, args, ae, max_grad_norm=0, model_file=None, eval_mode=False, **kwargs):
        super(ACAITrainer, self).__init__(args, ae, max_grad_norm, model_file, eval_mode, **kwargs)
        self.train_combined = "combined" in self.args["model"]
        self.dist_normal = None
        self.disc_model = None
        self._get_discriminator()
        kw = dict(lr=args["lr"], weight_decay=args.get("weight_decay", 0.0), betas=(0.9, 0.999))
        params = list(self.disc_model.parameters())
        if params and params[0].is_cuda and not self.eval_model:
            self.opt_disc = ops.HipAdam(params, on_step=self.disc_model.mark_weights_dirty, **kw)
        else:
            self.opt_disc = torch.optim.Adam(params, **kw)
        self.gamma_reg_acai = 0.2          # hyper-parameter of Berthelot et al., as in the reference
        self.z = None
        print("IMPORTANT --> {} training with combined losses: {}".format(self.__class__.__name__, self.train_combined))

    def _get_discriminator(self):
        from ..networks.acai_vanilla import Discriminator
        self.disc_model = Discriminator(self.args).to(self.args["device"])
        print("INFO - Trainer ACAI - Initiated discriminator")

    def _graph_ok(self, keep_predictions, eval_mode):
        return False                       # a fresh host-side alpha every step

    def _draw_alpha(self, B):
        """``torch.rand(B,1,1,1).to(device) / 2`` of the reference (:107): host generator, one draw per step."""
        return (torch.rand(B, 1, 1, 1) / 2).reshape(-1)

    def get_loss_disc(self, reconstruction, reference, z, is_test=True, out_mix=None, alpha=None):
        """Critic terms (reference :98-127).  ``out_mix`` / ``alpha``: the decoded acai mix and its coefficients when the caller
        already ran the decoder pass (training step); otherwise they are produced here."""
        dev = self.args["device"]
        B = z.size(0) // 2
        if alpha is None:
            alpha = self._draw_alpha(B)
        alpha = alpha.to(dev)
        ctxm = torch.no_grad() if is_test else torch.enable_grad()
        with ctxm:
            if out_mix is None:
                out_mix = self.model.decode(ops.lerp_mix(z, alpha, 1 - alpha))
            disc_mix_reg = reconstruction + self.gamma_reg_acai * (reference - reconstruction)
            d_reg, d_mix = self.disc_model.forward_multi([disc_mix_reg, out_mix])
            loss_disc_l2 = torch.mean(d_reg ** 2)
            loss_ae_l2 = torch.mean(d_mix ** 2)
            loss_disc_dist = torch.nn.functional.mse_loss(d_mix, alpha.reshape(-1), reduction="mean")
        return {"loss_disc_l2": loss_disc_l2, "loss_ae_l2": loss_ae_l2, "loss_disc_dist": loss_disc_dist}

    def train(self, batch_item, keep_predictions=True, eval_mode=False):
        if self.dp is not None and self.dp.active:
            raise NotImplementedError("ACAITrainer is single-process (the critic's gradients are not part of the data-parallel exchange)")
        dev = self.args["device"]
        x = self._to_device(batch_item["image"])
        between = self._to_device(batch_item["slice_between"])
        self._set_mode(not eval_mode)
        self._iters += 1
        B = x.shape[0] // 2
        if self.train_combined:
            z, z_ref = self.model.encode_multi([x, between], needs_grad=[True, False])
        else:
            z, z_ref = self.model.encode(x), None
        alpha = self._draw_alpha(B).to(dev)
        z_mix_acai = ops.lerp_mix(z, alpha, 1 - alpha)
        mask = batch_item["loss_mask"] if self.args.get("get_masks") else None
        if self.train_combined:
            a_from, a_to = self._mix_coefficients(batch_item, B)
            z_mix = ops.lerp_mix(z, a_from, a_to)
            out, out_mix, s_mix = self.model.decode_multi([z, z_mix_acai, z_mix])
        else:
            out, out_mix = self.model.decode_multi([z, z_mix_acai])
        loss_ae_dist = self.get_loss(x, out, is_test=False)["loss_ae_dist"]
        ld = self.get_loss_disc(out, x, z, is_test=False, out_mix=out_mix, alpha=alpha)
        loss_ae = loss_ae_dist + self.args["lamb_reg_acai"] * ld["loss_ae_l2"]
        loss_disc = ld["loss_disc_dist"] + ld["loss_disc_l2"]
        if self.train_combined:
            loss_latent = ops.mse_loss(z_mix.detach(), z_ref.detach())
            loss_ae = loss_ae + self.get_extra_loss(between, s_mix, z_mix, z=z, mask=mask, is_test=False)
        else:
            with torch.no_grad():
                r = self.synthesize_batch_images(batch_item=batch_item, z=z.detach(), compute_latent_loss=True, slice_between=between,
                                                 is_eval=True)
                self._set_mode(not eval_mode)
                s_mix, z_mix, loss_latent = r["s_between_mix"], r["z_mix"], r["loss_latent"]
                self.get_extra_loss(between, s_mix, z_mix, z=z, mask=mask, is_test=True)
        self.opt_ae.zero_grad()
        self.opt_disc.zero_grad()
        if not eval_mode:
            (loss_ae + loss_disc).backward()
            self.opt_ae.step()
            self.opt_disc.step()
        if self.opt_sched_ae is not None:
            self.opt_sched_ae.step()
        self._log("loss_ae", loss_ae)
        self._log("loss_latent_1", loss_latent)
        self._log("loss_disc", loss_disc)
        if keep_predictions:
            s = s_mix.detach().cpu()
            self.train_predictions = {"z_mix": z_mix.detach().cpu(), "pred_alphas": batch_item["alpha_from"].detach() if "alpha_from" in batch_item else torch.tensor([0.5]),
                                      "slice_inbetween_mix": s, "slice_inbetween_05": s, "reconstruction": out.detach().cpu()}

    def save_models(self, fname, epoch):
        if not self._is_writer():
            return

        def host(sd):
            return {k: (v.detach().cpu().contiguous() if torch.is_tensor(v) else v) for k, v in sd.items()}

        host_opt = optimizer_state_to_cpu         # a copy: never write into the dicts Optimizer.state_dict() hands out
        torch.save({"model_dict_ae": host(self.model.state_dict()), "optimizer_dict_ae": host_opt(self.opt_ae),
                    "model_disc": host(self.disc_model.state_dict()), "optimizer_disc": host_opt(self.opt_disc), "epoch": epoch}, fname)
