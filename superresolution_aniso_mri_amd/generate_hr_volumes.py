"""Slice-synthesis inference: the reference's ``generate_hr_volumes.py`` (:12-101 ``create_super_volume`` /
``latent_space_interp``, :104-183 I/O + ``main``) on the HIP engine.

The reference re-encodes both neighbour stacks for EVERY alpha (2n encoder passes per slice pair) and
copies every decoded stack to the host.  Here each slice is encoded ONCE (eval-mode BatchNorm makes results independent
of batch composition), the latents stay resident in HBM, the decoder's first convolution runs once per SLICE (it is linear: its
output for a latent mix is the mix of its outputs), all (z-1)*n mixes are formed on its pre-activations by ONE launch and the rest
of the decoder runs on them as ONE batch, the interleave and clamp happen on the device and there is a single device-to-host copy
at the end.
Conventions kept: ``alpha*enc(later slice) + (1-alpha)*enc(earlier slice)``, alphas = linspace(0,1,n+2)[1:-1],
output order [orig_0, interp_0(a_1..a_n), orig_1, ...], clamp to [0,1], new z-spacing = old/(n+1)."""
import argparse
import os
from pathlib import Path

import numpy as np
import torch

from . import _hip, ops


def _encode(trainer, x):
    return trainer.encode(x, use_sr_model=True)


def latent_space_interp(alpha, trainer, img1, img2, device=None, with_labels=False):
    """One alpha: decode(alpha*enc(img1) + (1-alpha)*enc(img2)) (reference :72-101).  Returns CPU tensors."""
    if with_labels:
        raise NotImplementedError("label channels (ACDCLBL multi-channel models) are outside this build")
    dev = device or trainer.args["device"]
    z = torch.cat([_encode(trainer, img1.float().to(dev)), _encode(trainer, img2.float().to(dev))], dim=0)
    inter = trainer.decode(ops.lerp_mix(z, float(alpha), float(1 - alpha)), use_sr_model=True)
    return {"inter_image": inter.detach().cpu().contiguous(), "inter_label": None}


def create_super_volume(trainer, images, alpha_range, use_original=False, labels=None, to_cpu=True):
    """images [z,1,y,x] or [z,y,x] -> {'upsampled_image': [(z-1)(n+1)+1, y, x] (CPU, clamped), 'upsampled_labels': None}.
    ``to_cpu=False`` leaves the result in HBM (callers that go on working on the device; bench.py times it that way)."""
    if labels is not None:
        raise NotImplementedError("label channels (ACDCLBL multi-channel models) are outside this build")
    if images.dim() == 3:
        images = torch.unsqueeze(images, dim=1)
    dev = trainer.args["device"]
    vol = images.float().to(dev)
    Z, _, H, W = vol.shape
    n = len(alpha_range)
    with torch.no_grad():
        lat = _encode(trainer, vol)                                   # every slice encoded exactly once
        recon = vol if use_original else trainer.decode(lat, use_sr_model=True)
        dec = None
        if Z > 1 and n > 0:
            model = trainer._use_sr_model(True)
            model.eval()
            dec = model.decode_mixes(lat, [float(a) for a in alpha_range]) if hasattr(model, "decode_mixes") else None
            if dec is None:
                zpair = torch.cat([lat[1:], lat[:-1]], dim=0)         # rows i / i+(Z-1): later slice / earlier slice
                mixes = torch.cat([ops.lerp_mix(zpair, float(a), float(1 - a)) for a in alpha_range], dim=0)
                dec = trainer.decode(mixes, use_sr_model=True)        # ONE decoder pass over all (Z-1)*n latents
        if vol.is_cuda and (H * W) % 4 == 0:
            out = ops.interleave_clamp(recon[:, 0], dec, n, 0.0, 1.0)        # interleave + clamp: one pass over the volume
        else:
            out = torch.empty(((Z - 1) * (n + 1) + 1, H, W), device=vol.device, dtype=torch.float32)
            out[::n + 1] = recon[:, 0]
            if dec is not None:
                dec = dec.reshape(n, Z - 1, H, W)
                for k in range(n):
                    out[k + 1::n + 1] = dec[k]
            out.clamp_(0, 1.)
    if to_cpu:
        out = out.cpu()
        if vol.is_cuda:
            _hip.check_device_watchdogs("create_super_volume")       # the volume leaves the device here: never a silent garbage volume
    return {"upsampled_image": out, "upsampled_labels": None}


# ---- I/O around the path (SimpleITK is optional; .npy volumes work everywhere) -----------------------------------------
def normalize_img(img, perc=(1, 99)):
    lo, hi = np.percentile(img, perc)
    return ((img.astype(img.dtype) - lo) / (hi - lo)).clip(0, 1)


def array_to_torch(np_img):
    np_img = np.asarray(np_img, dtype=np.float32)
    if np_img.max() > 1 or np_img.min() < 0:
        np_img = normalize_img(np_img)
    return torch.from_numpy(np_img).float().unsqueeze(dim=1)


def _sitk():
    try:
        import SimpleITK as sitk
        return sitk
    except ImportError:
        return None


def load_images(input_dir, suffix=".nii*"):
    """[(path, volume)] with volume a SimpleITK image (nii / mha / mhd) or a numpy array (.npy)."""
    input_dir, sitk = Path(input_dir), _sitk()
    files = []
    if sitk is not None:
        for pat in ("*" + suffix, "*.mha", "*.mhd"):
            files = sorted(input_dir.rglob(pat))
            if files:
                return [(f, sitk.ReadImage(str(f))) for f in files]
    else:                                   # no SimpleITK: the built-in NIfTI-1 / MetaImage reader (volume_io.py)
        from . import volume_io
        for pat in ("*" + suffix, "*.mha", "*.mhd"):
            files = sorted(f for f in input_dir.rglob(pat) if not str(f).endswith(".raw"))
            if files:
                return [(f, volume_io.read_volume(f)) for f in files]
    files = sorted(input_dir.rglob("*.npy"))
    if not files:
        raise FileNotFoundError("Error - no files found in {} with extensions nii, mha, mhd or npy".format(input_dir))
    return [(f, np.load(str(f))) for f in files]


def upsample_volume(trainer, vol_np, num_interpolations):
    """[z,y,x] or [t,z,y,x] numpy -> through-plane upsampled numpy with the same leading layout."""
    alpha_range = np.linspace(0, 1, num_interpolations + 2, endpoint=True)[1:-1]
    if vol_np.ndim == 3:
        return create_super_volume(trainer, array_to_torch(vol_np), alpha_range, use_original=True)["upsampled_image"].numpy()
    return np.stack([upsample_volume(trainer, v, num_interpolations) for v in vol_np])


def main(argv=None):
    p = argparse.ArgumentParser(description="Generate through-plane super-resolved volumes")
    p.add_argument("--exper_dir", type=str, default=None)
    p.add_argument("--model_nbr", type=int, default=None)
    p.add_argument("--num_interpolations", type=int, default=6)
    p.add_argument("--data_input_dir", type=str, default=None)
    p.add_argument("--output_dir", type=str, default=None)
    p.add_argument("--save", action="store_true")
    args = p.parse_args(argv)
    from .kwatsch.get_trainer import get_trainer_dynamic
    out_dir = Path(args.output_dir if args.output_dir is not None else os.path.join(args.exper_dir, "ni0{}".format(args.num_interpolations)))
    out_dir.mkdir(parents=True, exist_ok=True)
    images = load_images(Path(args.data_input_dir))
    print("INFO - Found {} files to process in {}".format(len(images), args.data_input_dir))
    trainer, _ = get_trainer_dynamic(src_path=args.exper_dir, model_nbr=args.model_nbr, model_nbr_sr=None, eval_mode=True)
    sitk = _sitk()
    results = []
    from . import volume_io
    for fname, img in images:
        if isinstance(img, volume_io.Volume):
            hr = upsample_volume(trainer, img.array, args.num_interpolations)
            spacing = list(img.spacing)
            spacing[2] = spacing[2] / (args.num_interpolations + 1)
            results.append((out_dir / fname.name, hr))
            if args.save:
                volume_io.write_volume(out_dir / fname.name, img, hr.astype(np.float32), spacing)
        elif isinstance(img, np.ndarray):
            hr = upsample_volume(trainer, img, args.num_interpolations)
            results.append((out_dir / fname.name, hr))
            if args.save:
                np.save(str(out_dir / fname.name), hr)
        else:
            arr = sitk.GetArrayFromImage(img)
            hr = upsample_volume(trainer, arr, args.num_interpolations)
            spacing = list(img.GetSpacing())
            zi = 2 if arr.ndim == 3 else 2
            spacing[zi] = spacing[zi] / (args.num_interpolations + 1)
            if hr.ndim == 4:
                out = sitk.JoinSeries([sitk.GetImageFromArray(v, False) for v in hr])
            else:
                out = sitk.GetImageFromArray(hr)
            out.SetOrigin(img.GetOrigin())
            out.SetDirection(img.GetDirection())
            out.SetSpacing(spacing)
            results.append((out_dir / fname.name, out))
            if args.save:
                sitk.WriteImage(out, str(out_dir / fname.name))
        print("Processed {}".format(fname))
    return results


if __name__ == "__main__":
    main()
