"""Oracle (TEST INFRASTRUCTURE, see oracle/__init__.py): the ``ae_combined`` training step and the
slice-synthesis inference loop, restated on the CPU oracle networks.

Follows
  * kwatsch/cardiac/trainer_ae.py:10-50 (AETrainerEndToEnd.train), :79-130 (extra loss), :165-182 (lerp)
  * kwatsch/brain/trainer_ae.py:92-132, :163-227, :255-281 (per-sample alpha_from / alpha_to)
  * kwatsch/base_trainer.py:164-198 (get_loss: MSE mean), :348-351 (0.5/0.5 mix)
  * kwatsch/trainer_ae.py:28-30 (Adam lr, betas (momentum|0.9, 0.999), weight_decay)
  * generate_hr_volumes.py:12-69, :72-101 (create_super_volume / latent_space_interp)
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import lpips_oracle


class OracleStep:
    """One ``ae_combined`` trainer on the oracle AE (device-agnostic CPU restatement).  ``recon_loss="perceptual"`` is
    ``--use_percept_loss`` (kwatsch/base_trainer.py:165-175: LPIPS(recons, reference) instead of the MSE); ``plain=True`` is the
    plain ``ae`` model (kwatsch/trainer_ae.py:71-109: reconstruction loss only)."""

    def __init__(self, ae, lr=1e-5, weight_decay=0.0, momentum=0.9, ex_loss_weight1=0.05,
                 image_mix_loss_func="perceptual", vgg_sd=None, lin_w=None, recon_loss="mse", plain=False):
        self.ae = ae
        self.opt = torch.optim.Adam(ae.parameters(), lr=lr, weight_decay=weight_decay, betas=(momentum, 0.999))
        self.lam = ex_loss_weight1
        self.mix_loss = image_mix_loss_func
        self.vgg_sd, self.lin_w = vgg_sd, lin_w
        self.recon_loss, self.plain = recon_loss, plain

    def extra_image_loss(self, reference, synthesized, route=None):
        # kwatsch/cardiac/trainer_ae.py:103-130 without masks / laploss.  The synthesised slice is the TARGET argument here (= in0 of the
        # distance network): its decisions are named ``lp_syn/in0/...`` (oracle/routing.py)
        if self.mix_loss == "perceptual":
            return lpips_oracle.perceptual_loss(reference, synthesized, self.vgg_sd, self.lin_w,
                                                normalize=True, route=route, tag="lp_syn/").mean()
        return F.mse_loss(reference, synthesized)

    def reconstruction_loss(self, reference, recons, route=None):
        # kwatsch/base_trainer.py:164-198: percept_criterion(recons, reference, normalize=True).mean() or F.mse_loss(recons, reference)
        # (the reconstruction is the PRED argument = in1: ``lp_rec/in1/...``)
        if self.recon_loss == "perceptual":
            return lpips_oracle.perceptual_loss(recons, reference, self.vgg_sd, self.lin_w, normalize=True, route=route, tag="lp_rec/").mean()
        return F.mse_loss(recons, reference, reduction="mean")

    def train_plain(self, image, slice_between, update=True):
        """kwatsch/trainer_ae.py:71-109.  The latent loss is for the log only: ``get_latent_loss(no_grad=True)`` encodes
        slice_between through ``self.encode(eval=True)`` (kwatsch/base_trainer.py:200-211,243-282), which switches the model to
        eval mode and nothing switches it back before ``_get_mixup_image`` (:337-346) -- the 0.5-mix is decoded in EVAL mode."""
        ae = self.ae
        B = image.shape[0] // 2
        z = ae.encode(image, train=True)
        out = ae.decode(z, train=True)
        loss = self.reconstruction_loss(image, out)
        with torch.no_grad():
            z_mix = 0.5 * z[:B] + 0.5 * z[B:]
            z_ref = ae.encode(slice_between, train=False)
            loss_latent = F.mse_loss(z_mix, z_ref)
        self.opt.zero_grad()
        if update:
            loss.backward()
            self.opt.step()
        with torch.no_grad():
            s_mix = ae.decode(z_mix, train=False)
        return dict(loss_ae=float(loss.detach()), loss_ae_dist=float(loss.detach()), loss_latent_1=float(loss_latent), z=z.detach(),
                    out=out.detach(), z_mix=z_mix.detach(), s_mix=s_mix)

    def train(self, image, slice_between, alpha_from=None, alpha_to=None, update=True, lam=None, route=None):
        """image [2B,1,H,W] (from-slices then to-slices), slice_between [B,1,H,W].
        alpha_* None -> cardiac 0.5/0.5 (trainer_ae.py:51, cardiac/trainer_ae.py:173);
        else [B,1] per-sample coefficients (brain/trainer_ae.py:264-266).  ``lam``: this step's synthesis-loss weight when loss
        annealing is on (kwatsch/cardiac/trainer_ae.py:80-83: ``loss_weights[epoch]``).  ``route`` (oracle/routing.py, optional):
        records / forces the non-smooth decisions of the differentiated passes -- ``x/enc.<i>``, ``z/dec.<i>``, ``mix/dec.<i>``,
        ``lp_rec/in1/relu<n>|pool<n>``, ``lp_syn/in0/...``; the default (None) runs the stock F.leaky_relu / relu / max_pool2d."""
        if self.plain:
            return self.train_plain(image, slice_between, update=update)
        ae = self.ae
        B = image.shape[0] // 2
        z = ae.encode(image, train=True, route=route, tag="x/")
        out = ae.decode(z, train=True, route=route, tag="z/")
        loss_ae_dist = self.reconstruction_loss(image, out, route=route)             # base_trainer.py:177
        if alpha_from is None:
            z_mix = 0.5 * z[:B] + (1 - 0.5) * z[B:]
        else:
            z_mix = alpha_from[:, :, None, None] * z[:B] + alpha_to[:, :, None, None] * z[B:]
        s_mix = ae.decode(z_mix, train=True, route=route, tag="mix/")
        z_ref = ae.encode(slice_between, train=True)       # graph never back-propagated; updates BN stats (Q6)
        loss_latent = F.mse_loss(z_mix, z_ref)
        loss_extra = (self.lam if lam is None else lam) * self.extra_image_loss(slice_between, s_mix, route=route)
        loss = loss_ae_dist + loss_extra
        self.opt.zero_grad()
        if update:
            loss.backward()
            self.opt.step()
        return dict(loss_ae=float(loss.detach()), loss_ae_dist=float(loss_ae_dist.detach()),
                    loss_ae_dist_extra=float(loss_extra.detach()), loss_latent_1=float(loss_latent.detach()), z=z.detach(), out=out.detach(), z_mix=z_mix.detach(),
                    s_mix=s_mix.detach())


def annealing_weights(epochs, ex_loss_weight1):
    """kwatsch/base_trainer.py:456-459: sigmoid(linspace(-5, 5, epochs)) * lambda, reversed; indexed by the trainer's epoch."""
    x = np.linspace(-5, 5, epochs)
    return (1.0 / (1.0 + np.exp(-x)) * ex_loss_weight1)[::-1].copy()


class OracleACAIStep:
    """kwatsch/trainer_acai.py:34-127 on the oracle networks: ``critic`` is an OracleAE whose ENCODER program/parameters are the
    Discriminator's (networks/acai_vanilla.py:140-153: encoder, then the mean over all latent elements per sample).
    Two Adam optimisers (:42-43), gamma = 0.2 (:44), both backward calls of :78-79 without detaching anything."""

    def __init__(self, ae, critic, lr=1e-5, weight_decay=0.0, momentum=0.9, lamb_reg_acai=0.5, ex_loss_weight1=0.001, combined=True,
                 image_mix_loss_func="mse", vgg_sd=None, lin_w=None):
        self.ae, self.critic = ae, critic
        self.opt = torch.optim.Adam(ae.parameters(), lr=lr, weight_decay=weight_decay, betas=(momentum, 0.999))
        self.opt_disc = torch.optim.Adam(critic.parameters(), lr=lr, weight_decay=weight_decay, betas=(0.9, 0.999))
        self.lamb, self.lam, self.combined, self.gamma = lamb_reg_acai, ex_loss_weight1, combined, 0.2
        self._extra = OracleStep(ae, lr=lr, ex_loss_weight1=ex_loss_weight1, image_mix_loss_func=image_mix_loss_func, vgg_sd=vgg_sd,
                                 lin_w=lin_w).extra_image_loss

    def disc(self, img):
        return self.critic.encode(img, train=True).reshape(img.shape[0], -1).mean(-1)

    def train(self, image, slice_between, alpha_from, alpha_to, alpha):
        """alpha: the [B] tensor the reference draws with torch.rand(B,1,1,1)/2 (:107)."""
        ae, B = self.ae, image.shape[0] // 2
        z = ae.encode(image, train=True)
        out = ae.decode(z, train=True)
        loss_ae_dist = F.mse_loss(out, image)
        loss_disc_l2 = torch.mean(self.disc(out + self.gamma * (image - out)) ** 2)                    # :100-105
        a4 = alpha.reshape(-1, 1, 1, 1)
        out_mix = ae.decode(a4 * z[:B] + (1 - a4) * z[B:], train=True)                                 # :108,:116
        disc_mix = self.disc(out_mix)
        loss_ae_l2 = torch.mean(disc_mix ** 2)
        loss_disc_dist = F.mse_loss(disc_mix, alpha.reshape(-1))
        loss_ae = loss_ae_dist + self.lamb * loss_ae_l2                                                # :64
        loss_disc = loss_disc_dist + loss_disc_l2                                                      # :65
        z_mix = alpha_from[:, :, None, None] * z[:B] + alpha_to[:, :, None, None] * z[B:]
        if self.combined:
            s_mix = ae.decode(z_mix, train=True)
            z_ref = ae.encode(slice_between, train=True)
            loss_extra = self.lam * self._extra(slice_between, s_mix)
            loss_ae = loss_ae + loss_extra
        else:
            with torch.no_grad():
                s_mix = ae.decode(z_mix, train=False)
                z_ref = ae.encode(slice_between, train=False)
                loss_extra = self.lam * self._extra(slice_between, s_mix)
        loss_latent = F.mse_loss(z_mix, z_ref)
        self.opt.zero_grad()
        self.opt_disc.zero_grad()
        loss_ae.backward(retain_graph=True)
        loss_disc.backward()
        self.opt.step()
        self.opt_disc.step()
        return dict(loss_ae=float(loss_ae.detach()), loss_disc=float(loss_disc.detach()), loss_ae_dist=float(loss_ae_dist.detach()),
                    loss_ae_dist_extra=float(loss_extra.detach()), loss_latent_1=float(loss_latent.detach()), out=out.detach(),
                    s_mix=s_mix.detach(), out_mix=out_mix.detach())


def create_super_volume(ae, images, alpha_range, use_original=True):
    """generate_hr_volumes.py:12-69 with the reference's per-alpha re-encoding collapsed (eval-mode BN
    makes results batch-composition independent).  images [z,1,H,W] -> [(z-1)(n+1)+1, H, W]."""
    with torch.no_grad():
        lat = ae.encode(images.float(), train=False)
        recon = images if use_original else ae.decode(lat, train=False)
        interp = []
        for alpha in alpha_range:
            inter = float(alpha) * lat[1:] + (1 - float(alpha)) * lat[:-1]   # :88, later slice weighted alpha
            interp.append(ae.decode(inter, train=False))
        interp = torch.cat(interp, dim=1)                                   # [z-1, n, H, W]
        parts = []
        for i in range(images.shape[0] - 1):
            parts += [recon[i], interp[i]]
        parts.append(recon[-1])
        return torch.clamp(torch.cat(parts, dim=0), min=0, max=1.)


def create_super_volume_eval(ae, images, alpha_range=None, use_original=True, downsample_steps=None,
                             generate_inbetween_slices=False):
    """evaluate/common.py:134-235, the evaluation protocol around the same synthesis: images [z,H,W].
    :156-169 keep every ``downsample_steps``-th slice (default len(alpha_range)+1 when in-between slices are generated) after
    cutting the (z-1) % steps remainder slices; :178-179 default alphas; :220-231 the remainder slices of the ORIGINAL
    volume are appended after the synthesised stack; pred_alphas = each alpha repeated for the z'-1 pairs (:206-207)."""
    if alpha_range is None:
        alpha_range = [0.25, 0.5, 0.75]
    if generate_inbetween_slices and downsample_steps is None:
        downsample_steps = int(len(alpha_range) + 1)
    orig = images
    rem = 0
    if downsample_steps is not None or generate_inbetween_slices:
        rem = (orig.shape[0] - 1) % downsample_steps
        if rem:
            images = images[:-rem]
        images = images[::downsample_steps]
    hr = create_super_volume(ae, images[:, None], alpha_range, use_original=use_original)
    if generate_inbetween_slices and rem:
        hr = torch.clamp(torch.cat([hr, orig[-rem:].float()]), min=0, max=1.)
    alphas = torch.tensor([float(a) for a in alpha_range], dtype=torch.float32).repeat_interleave(images.shape[0] - 1)
    return hr, alphas


def val_volume_compare(ae, image4d, frame_id, eval_patch_size, downsample_steps=2):
    """kwatsch/base_trainer.py:149-162 -> evaluate/evaluate_image.py:37-80,83-106 for ONE patient: frame ``frame_id`` of a [t,z,y,x]
    image is padded / centre-cropped to the evaluation patch (datasets/shared_transforms.py AdjustToPatchSize + CenterCrop), every 2nd
    slice kept and reconstructed, the held-out ones synthesised at alpha 0.5 (evaluate/common.py:134-235 with use_original=False,
    generate_inbetween_slices=True); returns (orig [z,y,x], synth [z',y,x], the [7k,1,y,x] stack handed to make_grid, k)."""
    from . import augment_oracle
    f = min(int(frame_id), image4d.shape[0] - 1)
    orig = torch.from_numpy(np.ascontiguousarray(augment_oracle.adjust_and_center_crop(np.asarray(image4d[f], dtype=np.float32), int(eval_patch_size))))
    synth, _ = create_super_volume_eval(ae, orig, [0.5], use_original=False, downsample_steps=downsample_steps, generate_inbetween_slices=True)
    real, syn = orig.numpy(), synth.numpy()
    if real.shape[0] % downsample_steps == 0:                     # evaluate_image.py:88-92
        real, syn = real[:-1], syn[:-1]
    n = real.shape[0]
    s_mask = np.ones(n, dtype=bool)
    s_mask[::downsample_steps] = False
    r_mask = ~s_mask
    s1, s3 = real[r_mask][:-1], real[r_mask][1:]
    r1, r3 = syn[r_mask][:-1], syn[r_mask][1:]
    held, made = real[s_mask], syn[s_mask]
    stack = np.concatenate([s1[:, None], r1[:, None], held[:, None], made[:, None], (held - made)[:, None], r3[:, None], s3[:, None]], axis=0)
    return orig.numpy(), synth.numpy(), stack, s1.shape[0]


def synthetic_triplets(B, H, W, seed, device="cpu"):
    """Smooth, correlated (from, to, between) triplets in [0,1]: sum of 8 Gaussian blobs
    + low-pass noise; between = 0.5(from+to) + N(0, 0.02)."""
    g = torch.Generator().manual_seed(int(seed))
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")

    def blobs(c, s, a):
        img = torch.zeros(B, H, W)
        for k in range(c.shape[1]):
            d = (yy[None] - c[:, k, 0, None, None]) ** 2 + (xx[None] - c[:, k, 1, None, None]) ** 2
            img = img + a[:, k, None, None] * torch.exp(-d / (2 * s[:, k, None, None] ** 2))
        return img

    c = torch.rand(B, 8, 2, generator=g)
    s = 0.04 + 0.12 * torch.rand(B, 8, generator=g)
    a = 0.2 + 0.5 * torch.rand(B, 8, generator=g)
    dc = 0.03 * torch.randn(B, 8, 2, generator=g)
    frm = blobs(c, s, a)
    to = blobs(c + dc, s, a)

    def lowpass(n):
        return F.avg_pool2d(F.pad(n[:, None], (2, 2, 2, 2), mode="reflect"), 5, stride=1)[:, 0]

    frm = (frm + 0.05 * lowpass(torch.randn(B, H, W, generator=g))).clamp(0, 1)
    to = (to + 0.05 * lowpass(torch.randn(B, H, W, generator=g))).clamp(0, 1)
    btw = (0.5 * (frm + to) + 0.02 * torch.randn(B, H, W, generator=g)).clamp(0, 1)
    image = torch.cat([frm[:, None], to[:, None]], dim=0).float().to(device)
    return image, btw[:, None].float().to(device)


def ssim(a, b, data_range=1.0, win=7, k1=0.01, k2=0.03):
    """Mean SSIM, uniform win x win window, sample covariance (skimage ``structural_similarity`` defaults for
    float images with explicit data_range; evaluate/metrics.py:139).  a, b: [N,1,H,W] or [H,W]."""
    a = torch.as_tensor(a, dtype=torch.float64).reshape(-1, 1, *a.shape[-2:])
    b = torch.as_tensor(b, dtype=torch.float64).reshape(-1, 1, *b.shape[-2:])
    npx = win * win
    cov_norm = npx / (npx - 1.0)
    ux, uy = F.avg_pool2d(a, win, 1), F.avg_pool2d(b, win, 1)
    uxx, uyy, uxy = F.avg_pool2d(a * a, win, 1), F.avg_pool2d(b * b, win, 1), F.avg_pool2d(a * b, win, 1)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux ** 2 + uy ** 2 + c1) * (vx + vy + c2))
    return float(s.mean())


def psnr(a, b, data_range=1.0):
    mse = float(((torch.as_tensor(a, dtype=torch.float64) - torch.as_tensor(b, dtype=torch.float64)) ** 2).mean())
    return 10 * np.log10(data_range ** 2 / max(mse, 1e-30))
