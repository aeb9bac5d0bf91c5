"""TEST INFRASTRUCTURE: numpy restatement of the reference's training-time sample pipeline for ae_combined on ACDC
(train_cardiac_aesr.py:83-96): AdjustToPatchSize -> CenterCrop -> RandomCrop -> RandomIntensity -> RandomRotation on one
[3,H,W] triplet (from, to, between), datasets/shared_transforms.py:48-120,224-254,297-363,366-386,389-447, and the batch
layout of prepare_batch_pairs (datasets/ACDC/data4d_simple.py:327-355).  Random numbers are drawn from the caller's
numpy RandomState in the reference's order: [top, left] (only if a crop happens), gain, cutoff, k."""
import numpy as np


def adjust_and_center_crop(img, aug):
    """[3,H,W] -> [3,aug',aug']: zero-pad up to aug (shared_transforms.py:389-447), then centre crop (:297-363)."""
    _, w, h = img.shape                       # the reference calls dim 1 "w" and dim 2 "h" here
    dwl = dwr = dhl = dhr = 0
    if w < aug:
        d = aug - w
        dwl, dwr = d // 2, (d // 2 if d % 2 == 0 else d // 2 + 1)
    if h < aug:
        d = aug - h
        dhl, dhr = d // 2, (d // 2 if d % 2 == 0 else d // 2 + 1)
    img = np.pad(img, ((0, 0), (dwl, dwr), (dhl, dhr)), "constant", constant_values=(0,)).astype(np.float32)
    _, hh, ww = img.shape
    half = int(aug / 2)
    sw = slice(int(ww / 2) - half, int(ww / 2) + half)
    sh = slice(int(hh / 2) - half, int(hh / 2) + half)
    return img[:, sh, sw]


def draw_params(rs, h, w, width):
    """The random numbers of one sample in the reference's order."""
    top = left = 0
    if not (h == width and w == width):
        top = int(rs.randint(0, h - width))
        left = int(rs.randint(0, w - width))
    gain = float(rs.uniform(2.5, 7.5))
    cutoff = float(rs.uniform(0.25, 0.75))
    k = int(rs.randint(0, 4))
    return top, left, gain, cutoff, k


def augment_triplet(triplet, aug, width, rs):
    """One sample: [3,H,W] float32 -> ([3,width,width] float32, params)."""
    img = adjust_and_center_crop(np.asarray(triplet, dtype=np.float32), aug)
    _, h, w = img.shape
    top, left, gain, cutoff, k = draw_params(rs, h, w, width)
    img = img[:, top:top + width, left:left + width]
    img = (1 / (1 + np.exp(gain * (cutoff - img)))).astype(np.float32)
    img = np.rot90(img, k, (1, 2)).copy()
    return img, (top, left, gain, cutoff, k)


def assemble_batch(samples):
    """List of [3,W,W] -> {'image': [2B,1,W,W] (all 'from' slices, then all 'to' slices), 'slice_between': [B,1,W,W]}."""
    a = np.stack([s[0] for s in samples])[:, None]
    b = np.stack([s[1] for s in samples])[:, None]
    m = np.stack([s[2] for s in samples])[:, None]
    return {"image": np.concatenate([a, b], axis=0), "slice_between": m}
