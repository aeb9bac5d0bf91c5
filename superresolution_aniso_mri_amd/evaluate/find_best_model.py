"""Model selection over the checkpoints of an experiment: the reference's ``evaluate/find_best_model.py`` (:43-110
``find_best_val_model``, :31-37 ``store_top_scores``, :113-140 ``load_model_scores``) with the evaluation it drives
(``evaluate/evaluate_interpolations.py:45-63``, ``evaluate/create_HR_images.py:121-178,240-420``) on the HIP path.

Per checkpoint and volume: pad / centre-crop to ``ps_evaluate`` (AdjustToPatchSize + CenterCrop,
``datasets/shared_transforms.py:297-363,389-447``), keep every ``downsample_steps``-th slice, synthesise the slices in between
(``evaluate.common.create_super_volume(generate_inbetween_slices=True, use_original=False)``), score the volume up to the last
paired slice against the original: all slices, the synthesised ones (s_mask) and the reconstructed ones (r_mask) --
SSIM and PSNR from ONE device pass per volume (``evaluate.metrics.slice_ssim_psnr``), VIF from another
(``evaluate.metrics.slice_vif``: the reference's ``vifp_mscale`` on uint8 slices, evaluate/vifvec.py:7-63); LPIPS per slice is off, as in
the reference's call (``compute_percept_loss = False``)."""
import glob
import os
import types

import numpy as np
import torch

from . import common as _common
from . import metrics as _metrics


def adjust_and_center_crop(image, patch_size):
    """[z,y,x] (or [y,x]) numpy -> zero-padded to at least ``patch_size`` per side, then centre-cropped to it.  Keeps the
    reference's padding rule, which derives BOTH paddings from patch_size[0] (shared_transforms.py:433-445)."""
    ps = (patch_size, patch_size) if not isinstance(patch_size, tuple) else patch_size
    image = np.asarray(image, dtype=np.float32)
    w, h = image.shape[-2:]

    def deltas(n, target):
        if n >= target:
            return 0, 0
        d = ps[0] - n
        return d // 2, (d // 2 if d % 2 == 0 else d // 2 + 1)
    pw, ph = deltas(w, ps[0]), deltas(h, ps[1])
    pad = [(0, 0)] * (image.ndim - 2) + [pw, ph]
    image = np.pad(image, pad, "constant", constant_values=(0,)).astype(np.float32)
    hh, ww = image.shape[-2:]                       # CenterCrop :316-333 (its "h, w" are the last two axes)
    half_w, half_h = int(ps[0] / 2), int(ps[1] / 2)
    sl_w = slice(int(ww / 2) - half_w, int(ww / 2) + half_w)
    sl_h = slice(int(hh / 2) - half_h, int(hh / 2) + half_h)
    return image[..., sl_h, sl_w]


def get_transforms(transform_patch_size, to_tensor=True):
    """Callable on the reference's sample dicts ({'image': ...}); ``to_tensor`` converts the image to a torch tensor."""
    def transform(sample):
        out = dict(sample)
        out["image"] = adjust_and_center_crop(sample["image"], transform_patch_size)
        if to_tensor:
            out["image"] = torch.from_numpy(np.ascontiguousarray(out["image"]))
        return out
    return transform


def generate_synth_slices_mask(orig_num_slices, downsample_steps):
    """(r_mask, s_mask) over the slices up to the last paired one: reconstructed originals / synthesised in-betweens
    (evaluate/quantitative_comparison.py:10-17)."""
    n = _common.determine_last_slice(orig_num_slices, downsample_steps) + 1
    s_mask = np.ones(n, dtype=bool)
    s_mask[::downsample_steps] = False
    return ~s_mask, s_mask


def compute_metrics(images_ref, new_images, downsample_steps, data_range=1.0, compute_percept_loss=False, percept_loss=None):
    """{'ssim','psnr','vif' and their '_synth' / '_recon' forms [,'lpips']} of one volume (create_HR_images.py:121-178; LPIPS over all
    scored slices only, as there: the masked calls pass compute_percept_loss=False).  VIF: mean over the slices of the selection whose
    score is finite (evaluate/metrics.py:100-106)."""
    last = _common.determine_last_slice(images_ref.shape[0], downsample_steps) + 1
    r_mask, s_mask = generate_synth_slices_mask(images_ref.shape[0], downsample_steps)
    ssim, psnr, _ = _metrics.slice_ssim_psnr(images_ref[:last], new_images[:last], data_range=data_range)

    def mean_psnr(sel):
        v = psnr[sel]
        return float(np.mean(v[np.isfinite(v)]))
    vif = _metrics.slice_vif(images_ref[:last], new_images[:last])

    def mean_vif(sel):
        v = vif[sel]
        v = v[np.isfinite(v)]
        return float(np.mean(v)) if v.size else float("nan")
    everything = np.ones(last, dtype=bool)
    out = {"ssim": float(np.mean(ssim)), "psnr": mean_psnr(everything), "vif": mean_vif(everything),
           "ssim_synth": float(np.mean(ssim[s_mask])), "psnr_synth": mean_psnr(s_mask), "vif_synth": mean_vif(s_mask),
           "ssim_recon": float(np.mean(ssim[r_mask])), "psnr_recon": mean_psnr(r_mask), "vif_recon": mean_vif(r_mask)}
    if compute_percept_loss:
        out["lpips"] = _metrics.compute_lpips_for_batch(images_ref[:last], new_images[:last], criterion=percept_loss)
    return out


def _as_list(data_generator):
    if isinstance(data_generator, dict):
        return list(data_generator.values())
    return list(data_generator)


def evaluate_interpolation_performance(trainer, myargs, data_generator, transform=None, downsample_steps=None, file_suffix=None,
                                       patient_id=None, eval_axis=0, compute_percept_loss=False, percept_loss=None):
    """evaluate/evaluate_interpolations.py:45-63 -> create_hr_images(generate_inbetween_slices=True, use_original_slice=False,
    num_interpolations = downsample_steps - 1): result lists per volume."""
    if eval_axis != 0:
        raise NotImplementedError("long-axis (eval_axis != 0) evaluation is outside the ae_combined path")
    alpha_range = np.linspace(0, 1, (downsample_steps - 1) + 2, endpoint=True)[1:-1]
    keys = ("ssim", "psnr", "vif", "ssim_synth", "psnr_synth", "vif_synth", "ssim_recon", "psnr_recon", "vif_recon")
    res = {k: [] for k in keys + ("lpips", "lpips_synth", "lpips_recon")}
    for batch in _as_list(data_generator):
        if transform is not None:
            batch = transform(batch)
        pat = batch.get("patient_id")
        if patient_id is not None and str(pat) != str(patient_id):
            continue
        images = batch["image"]
        images = torch.from_numpy(np.ascontiguousarray(images)) if isinstance(images, np.ndarray) else images
        ref = batch.get("image_hr")
        out = _common.create_super_volume(trainer, images, alpha_range=alpha_range, use_original=False,
                                          downsample_steps=downsample_steps, generate_inbetween_slices=True)
        m = compute_metrics(images if ref is None else ref, out["upsampled_image"], downsample_steps,
                            compute_percept_loss=compute_percept_loss, percept_loss=percept_loss)
        for k in keys:
            res[k].append(m[k])
        if compute_percept_loss:
            res["lpips"].append(m["lpips"])
    return res


def store_top_scores(model_nbr, top_scores, ssim_results, psnr_results, vif_results):
    top_scores[model_nbr] = np.array([np.mean(np.array(ssim_results)), np.mean(np.array(psnr_results)), np.mean(np.array(vif_results))])
    return top_scores


def find_best_val_model(data_generator, exper_src_dir, epoch_range=None, ps_evaluate=None, eval_axis=0, downsample_steps=None,
                        patient_id=None, limit_4d=False, func_get_trainer=None):
    """Scores every ``<exper_src_dir>/models/<epoch>.models`` of ``epoch_range``; writes ``model_perf_<a>_to_<b>_axis<k>.npz``
    (all slices) and ``model_perf_synth_...npz`` (synthesised slices only) as the reference does; returns {epoch: [ssim, psnr, vif]}."""
    if func_get_trainer is None:
        from ..kwatsch.get_trainer import get_trainer_dynamic as func_get_trainer
    exper_src_dir = os.path.expanduser(exper_src_dir)
    search_mask = os.path.join(os.path.join(exper_src_dir, "models"), "*.models")
    model_list = sorted(glob.glob(search_mask))
    if epoch_range is not None:
        epoch_range = [str(e) for e in epoch_range]
        model_list = sorted(m for m in model_list if os.path.basename(m).replace(".models", "") in epoch_range)
    else:
        epoch_range = [os.path.basename(m).replace(".models", "") for m in model_list]
    print("INFO - find-best-validation-model - testing {} networks using p-size {}  - eval_axis={}".format(len(model_list), ps_evaluate,
                                                                                                         eval_axis))
    if len(model_list) == 0:
        raise ValueError("Error no models found with search mask {}".format(search_mask))
    if isinstance(data_generator, types.GeneratorType):
        items = list(data_generator)
        if limit_4d:
            items = [t for t in items if t.get("frame_id") in [4, 11, 15]]
        data_generator = {i: t for i, t in enumerate(items)}
    top_scores, top_synth = {}, {}
    best = {"ssim": (None, 0), "psnr": (None, 0), "ssim_synth": (None, 0), "psnr_synth": (None, 0)}
    transform = None if ps_evaluate is None else get_transforms(ps_evaluate, to_tensor=False)
    for model_nbr in epoch_range:
        trainer, e_args = func_get_trainer(src_path=exper_src_dir, model_nbr=model_nbr, eval_mode=True)
        if downsample_steps is None:
            if "downsample_steps" not in e_args:
                raise ValueError("ERROR - Downsample steps need to be specified")
            downsample_steps = e_args["downsample_steps"]
        r = evaluate_interpolation_performance(trainer, e_args, data_generator, transform=transform, downsample_steps=downsample_steps,
                                               patient_id=patient_id, eval_axis=eval_axis)
        top_scores = store_top_scores(model_nbr, top_scores, r["ssim"], r["psnr"], r["vif"])
        top_synth = store_top_scores(model_nbr, top_synth, r["ssim_synth"], r["psnr_synth"], r["vif_synth"])
        for key, scores, col in (("ssim", top_scores, 0), ("psnr", top_scores, 1), ("ssim_synth", top_synth, 0), ("psnr_synth", top_synth, 1)):
            if scores[model_nbr][col] > best[key][1]:
                best[key] = (int(model_nbr), float(scores[model_nbr][col]))
        del trainer
    print("Top metrics: Mean SSIM/PSNR M-{}: {:.4f} / M-{}: {:.4f}".format(best["ssim"][0], best["ssim"][1], best["psnr"][0], best["psnr"][1]))
    print("Top synthesis: Mean SSIM/PSNR M-{}: {:.4f} / M-{}: {:.4f}".format(best["ssim_synth"][0], best["ssim_synth"][1],
                                                                             best["psnr_synth"][0], best["psnr_synth"][1]))
    tag = "{}_to_{}_axis{}.npz".format(epoch_range[0], epoch_range[-1], eval_axis)
    np.savez(os.path.join(exper_src_dir, "model_perf_" + tag), **top_scores)
    np.savez(os.path.join(exper_src_dir, "model_perf_synth_" + tag), **top_synth)
    print("Saved result dict to {}".format(os.path.join(exper_src_dir, "model_perf_" + tag)))
    return dict(sorted(top_scores.items()))


def load_model_scores(exper_dir, file_suffix=".npz", synthesis=False):
    """(results, epochs, ssim, psnr, vif) from the files written by ``find_best_val_model`` (reference :113-140)."""
    load_dir = os.path.expanduser(exper_dir)
    files = glob.glob(os.path.join(load_dir, ("model_perf_synth*" if synthesis else "model_perf*") + file_suffix))
    if len(files) == 0:
        print("INFO - nothing to load from {}".format(load_dir))
        return None
    results = {}
    for fname in files:
        if not synthesis and "synth" in fname:
            continue
        f = np.load(fname)
        results.update({epoch: f[epoch] for epoch in f.files})
    epochs = [int(e) for e in results]
    m = np.array([results[str(e)] for e in epochs]).reshape(len(epochs), 3)
    return results, np.array(epochs), m[:, 0], m[:, 1], m[:, 2]
