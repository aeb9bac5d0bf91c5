"""Latent interpolation helpers + small image-grid utilities (reference kwatsch/acai_utils.py:26-103, with the
torchvision / imageio dependencies replaced by a few lines of numpy)."""
import numpy as np
import torch

from .. import ops


def clip_grad_norm(optimizer, max_norm, norm_type=2):
    for group in optimizer.param_groups:
        torch.nn.utils.clip_grad_norm_(group["params"], max_norm, norm_type)


def make_grid(t, nrow, padding=2, pad_value=0.5):
    """[N,C,H,W] -> [C, rows*(H+p)+p, cols*(W+p)+p] (the layout of torchvision.utils.make_grid)."""
    t = torch.as_tensor(t).detach().cpu().float()
    n, c, h, w = t.shape
    cols = min(nrow, n)
    rows = int(np.ceil(n / cols))
    grid = torch.full((c, rows * (h + padding) + padding, cols * (w + padding) + padding), float(pad_value))
    for k in range(n):
        r, q = divmod(k, cols)
        grid[:, padding + r * (h + padding): padding + r * (h + padding) + h,
             padding + q * (w + padding): padding + q * (w + padding) + w] = t[k]
    return grid


def generate_recon_grid(img_ref, img_recons, max_items=16):
    img_ref, img_recons = img_ref.detach().cpu().float(), img_recons.detach().cpu().float()
    k = min(max_items, img_recons.size(0))
    parts = torch.cat([img_ref[:k], img_recons[:k], img_ref[:k] - img_recons[:k]], dim=0)
    return make_grid(parts, k, padding=2, pad_value=0.5).numpy()


def generate_batch_compare_grid(batch_item, slice_inbetween_mix, reconstruction, max_items=8):
    ref = batch_item["slice_between"].detach().cpu().float()
    k = min(max_items, ref.size(0))
    return make_grid(torch.cat([ref[:k], slice_inbetween_mix[:k].float(), reconstruction[:k].float()], dim=0), k).numpy()


def save_image_grid(grid, filename):
    """Write a [C,H,W] grid in [0,1] as PNG when matplotlib is importable (visualisation is optional)."""
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        plt.imsave(filename, np.clip(np.asarray(grid)[0], 0, 1), cmap="gray")
    except Exception as e:      # pragma: no cover - optional dependency
        print("WARNING - could not write {}: {}".format(filename, e))


def interpolate_latents(trainer, z_a, z_b, num_interpol):
    """All interior points of the segment z_a -> z_b, decoded: t in linspace(0,1,n+2)[1:-1], z = a*(1-t) + b*t
    (reference :56,95).  One fused lerp launch per t; latents stay on the device."""
    ts = np.linspace(0., 1., num_interpol + 2)[1:-1]
    zcat = torch.cat([z_a, z_b], dim=0)
    mixes = [ops.lerp_mix(zcat, float(1 - t), float(t)) for t in ts]
    return trainer.decode(torch.cat(mixes, dim=0), eval=True, use_sr_model=True)


def interpolate_2(trainer, x, num_interpol=9, show_critic=False, side=None, clear_cache=False, eval=True):
    """Grid of [first half | interpolations | second half] (reference :41-79)."""
    if show_critic:
        raise NotImplementedError("show_critic (the critic's score row under the grid, reference :59-62,76-78) is a visualisation option that is not built; "
                                  "the ACAI critic itself is: kwatsch/trainer_acai.py")
    side = x.size(0) // 2 if side is None else side
    z = trainer.encode(x, eval=eval)
    x_interp = interpolate_latents(trainer, z[:side], z[-side:], num_interpol)
    xc = x.detach().cpu().float()
    allimg = torch.cat([xc[:side], x_interp.detach().cpu().float().contiguous(), xc[-side:]], dim=0)
    return make_grid(allimg, side, padding=2, pad_value=0.5).numpy().squeeze().transpose(1, 2, 0) if allimg.shape[1] > 1 \
        else make_grid(allimg, side, padding=2, pad_value=0.5).numpy()[0]


def create_interpol_grid(trainer, x, num_interpol=9, slice_step=1):
    """Interpolate between every slice and its ``slice_step`` neighbour (reference :82-103)."""
    if x.dim() == 3:
        x = x[:, None]
    z = trainer.encode(x, eval=False) if trainer.model.training else trainer.encode(x, eval=True)
    x_interp = interpolate_latents(trainer, z[slice_step:], z[:-slice_step], num_interpol)
    xc = x.detach().cpu().float()
    allimg = torch.cat([xc[slice_step:], x_interp.detach().cpu().float().contiguous(), xc[:-slice_step]], dim=0)
    return make_grid(allimg, z.shape[0] - slice_step, padding=2, pad_value=0.5).numpy()[0]
