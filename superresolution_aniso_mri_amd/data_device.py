"""On-device batch assembly + augmentation for ae_combined training.

The reference feeds the step from a 2-worker numpy DataLoader: per sample a [3,H,W] triplet (from, to, between) is gathered
from a 4-D volume (datasets/ACDC/data4d_simple.py:191-240), padded / centre-cropped to ``aug_patch_size``, randomly cropped
to ``width``, passed through a random sigmoid intensity curve and rotated by a random multiple of 90 degrees
(train_cardiac_aesr.py:83-96; datasets/shared_transforms.py), then collated and re-laid out by ``prepare_batch_pairs``.
At several thousand slices per second that pipeline is the bottleneck.  Here the volumes live in HBM and ONE kernel launch
(``aesr_triplet_assemble``) produces the step's ``image`` [2B,1,W,W] / ``slice_between`` [B,1,W,W] tensors.  The random
numbers are drawn on the host from a numpy ``RandomState`` in the reference's order, so a single-worker reference loader with
the same state yields the same batch."""
import numpy as np
import torch

from . import _hip
from ._hip import check, lib, ptr, stream


def get_random_adjacent_slice(slice_id, num_slices, rs, step=1):
    """datasets/common.py:34-43"""
    last = num_slices - 1
    if slice_id + step > last:
        return slice_id - step
    if slice_id == 0:
        return step
    if slice_id - step < 0:
        return slice_id + step
    # rs.choice([a, b]) draws ONE rs.randint(0, 2) (numpy legacy RandomState.choice without p): the same stream position and result
    # at a third of the host time (8.6 -> 2.7 us per draw; the training loop makes ~10 draws per sample)
    return (slice_id - step, slice_id + step)[int(rs.randint(0, 2))]


def rescale_intensities(vol, percs=(1, 99)):
    """datasets/ACDC/data.py:151-160: percentile window -> [0, 1]."""
    lo, hi = np.percentile(vol, percs)
    lo = 0 if np.isnan(lo) else lo
    hi = 1 if np.isnan(hi) else hi
    return ((vol.astype(np.float32) - lo) / (hi - lo)).clip(0, 1)


def load_volume_dir(path):
    """All volumes of a directory (.npy, .nii, .nii.gz, .mha, .mhd; 3-D [Z,H,W] or 4-D [T,Z,H,W] -> one volume per frame),
    each rescaled to [0,1] by its 1st / 99th percentile when it is not already in that range."""
    import os
    from . import volume_io
    vols = []
    for name in sorted(os.listdir(path)):
        f = os.path.join(path, name)
        low = name.lower()
        if low.endswith(".npy"):
            arr = np.load(f)
        elif low.endswith((".nii", ".nii.gz", ".mha", ".mhd")):
            arr = volume_io.read_volume(f).array
        else:
            continue
        frames = [arr] if arr.ndim == 3 else list(arr)
        for v in frames:
            v = np.asarray(v, dtype=np.float32)
            if v.ndim != 3:
                raise ValueError("%s: expected a 3-D or 4-D volume, got shape %s" % (f, arr.shape))
            vols.append(v if (v.min() >= 0 and v.max() <= 1) else rescale_intensities(v))
    if not vols:
        raise FileNotFoundError("no .npy / .nii / .mha / .mhd volumes in %s" % path)
    return vols


def load_image_dict(path, max_patients=2):
    """The in-memory validation images ``validate(image_dict=...)`` previews (train_cardiac_aesr.py:49-53 of the reference: two 4-D
    patients): {p_id: {'image': [t,z,y,x] float32 in [0,1], 'patient_id': 'patientNNN', 'spacing': (z,y,x)}} from the first
    ``max_patients`` volumes of a directory (a 3-D volume counts as one frame).  p_id: the digits in the file name, else its rank."""
    import os
    import re
    from . import volume_io
    out = {}
    for rank, name in enumerate(sorted(os.listdir(path))):
        f, low = os.path.join(path, name), name.lower()
        if low.endswith(".npy"):
            arr, spacing = np.load(f), (1.0, 1.0, 1.0)
        elif low.endswith((".nii", ".nii.gz", ".mha", ".mhd")):
            v = volume_io.read_volume(f)
            arr, spacing = v.array, tuple(v.spacing[:3][::-1])
        else:
            continue
        arr = np.asarray(arr, dtype=np.float32)
        arr = arr[None] if arr.ndim == 3 else arr
        if arr.ndim != 4:
            raise ValueError("%s: expected a 3-D or 4-D image, got shape %s" % (f, arr.shape))
        if not (arr.min() >= 0 and arr.max() <= 1):
            arr = np.stack([rescale_intensities(a) for a in arr])
        digits = re.findall(r"\d+", name)
        p_id = int(digits[0]) if digits and int(digits[0]) not in out else 1000 + rank
        out[p_id] = {"image": arr, "patient_id": "patient{:03d}".format(p_id), "spacing": np.asarray(spacing, dtype=np.float64)}
        if len(out) >= int(max_patients):
            break
    if not out:
        raise FileNotFoundError("no .npy / .nii / .mha / .mhd images in %s" % path)
    return out


class TripletAugmenter:
    """``volumes``: list of float32 arrays [Z,H,W] already intensity-normalised to [0,1] (one per patient / frame)."""

    def __init__(self, volumes, width, aug_patch_size, rs=None, device="cuda"):
        self.width, self.aug = int(width), int(aug_patch_size)
        self.rs = rs if rs is not None else np.random.RandomState(1234)
        self.device = device
        offs, flat, self.shapes = [], [], []
        n = 0
        for v in volumes:
            v = np.ascontiguousarray(v, dtype=np.float32)
            if v.ndim != 3:
                raise ValueError("every volume must be [Z,H,W], got %s" % (v.shape,))
            offs.append(n)
            self.shapes.append(v.shape)
            flat.append(v.reshape(-1))
            n += v.size
        self.offsets = offs
        self.cache = torch.from_numpy(np.concatenate(flat)).to(device)        # the device-resident volume cache
        self._out = {}          # B -> (joined [3B,1,W,W] output buffer, alpha_from, alpha_to): reused by every batch of that size

    # ---- host-side random draws, in the reference's order ---------------------------------------------------------
    def _geometry(self, H, W):
        """Offset of the centre-cropped, padded image inside the source slice, and its size (shared_transforms.py:297-447)."""
        aug = self.aug
        dl_y = (aug - H) // 2 if H < aug else 0
        dl_x = (aug - W) // 2 if W < aug else 0
        Hp, Wp = max(H, aug), max(W, aug)
        half = int(aug / 2)
        sy, sx = int(Hp / 2) - half, int(Wp / 2) - half
        return sy - dl_y, sx - dl_x, 2 * half, 2 * half

    def draw_triplet(self, vol_id, slice_id, step=1):
        """Slice choice of the dataset's __getitem__ (data4d_simple.py:191-203): a random neighbour, the slice in between and
        a random from/to order.  With step == 1 the in-between slice is the neighbour itself in the reference; callers that
        train on sub-sampled volumes pass step = 2."""
        Z = self.shapes[vol_id][0]
        other = get_random_adjacent_slice(slice_id, Z, self.rs, step)
        between = (slice_id + other) // 2
        if int(self.rs.randint(0, 2)) == 0:            # == rs.choice([0, 1]), see get_random_adjacent_slice
            return slice_id, other, between
        return other, slice_id, between

    def draw_transform(self, vol_id):
        """[top, left] (only when a crop happens), gain, cutoff, k: shared_transforms.py:90-91, 376-377, 235."""
        _, H, W = self.shapes[vol_id]
        oy, ox, h, w = self._geometry(H, W)
        top = left = 0
        if not (h == self.width and w == self.width):
            top = int(self.rs.randint(0, h - self.width))
            left = int(self.rs.randint(0, w - self.width))
        gain = float(self.rs.uniform(2.5, 7.5))
        cutoff = float(self.rs.uniform(0.25, 0.75))
        k = int(self.rs.randint(0, 4))
        return oy + top, ox + left, gain, cutoff, k

    # ---- device side ------------------------------------------------------------------------------------------------
    def assemble(self, triplets, transforms=None, reuse_output=False):
        """triplets: list of (vol_id, z_from, z_to, z_between); transforms: list of (oy, ox, gain, cutoff, k) or None (drawn
        now, one sample after the other).  Returns {'image': [2B,1,W,W], 'slice_between': [B,1,W,W], 'alpha_from', 'alpha_to'}."""
        B, Wd = len(triplets), self.width
        if transforms is None:
            transforms = [self.draw_transform(t[0]) for t in triplets]
        if reuse_output:
            # ONE buffer [image | slice_between] per batch size, written again by every batch: the trainer's captured step reads its
            # inputs from exactly these addresses (AEBaseTrainer._train_graphed adopts "_persistent" batches as its static inputs), so
            # a training step costs the assemble launch and the graph replay -- no allocation, no fill, no copy of the batch
            if B not in self._out:
                both = torch.empty((3 * B, 1, Wd, Wd), device=self.device, dtype=torch.float32)
                half = torch.full((B, 1), 0.5, device=self.device, dtype=torch.float32)
                self._out[B] = (both, half, half.clone())
            both, a_from, a_to = self._out[B]
            image, between = both[:2 * B], both[2 * B:]
        else:
            image = torch.empty((2 * B, 1, Wd, Wd), device=self.device, dtype=torch.float32)
            between = torch.empty((B, 1, Wd, Wd), device=self.device, dtype=torch.float32)
        _hip.require_gpu_tensor(self.cache, "volume cache")
        for b0 in range(0, B, 64):
            n = min(64, B - b0)
            descs = (_hip.TripletDesc * n)()
            for i in range(n):
                vid, zf, zt, zb = triplets[b0 + i]
                oy, ox, gain, cutoff, k = transforms[b0 + i]
                Z, H, W = self.shapes[vid]
                if not (0 <= zf < Z and 0 <= zt < Z and 0 <= zb < Z):
                    raise ValueError("slice index outside volume %d (Z=%d)" % (vid, Z))
                descs[i] = _hip.TripletDesc(self.offsets[vid], H, W, zf, zt, zb, oy, ox, k, gain, cutoff)
            if B <= 64:
                img_dst, btw_dst = image, between
            else:       # more than one launch: each launch owns a contiguous [from | to] pair block -> assemble then scatter
                img_dst = torch.empty((2 * n, 1, Wd, Wd), device=self.device, dtype=torch.float32)
                btw_dst = between[b0:b0 + n]
            check(lib.aesr_triplet_assemble(ptr(self.cache), descs, n, Wd, ptr(img_dst), ptr(btw_dst), stream()),
                  "aesr_triplet_assemble")
            if B > 64:
                image[b0:b0 + n] = img_dst[:n]
                image[B + b0:B + b0 + n] = img_dst[n:]
        if reuse_output:
            return {"image": image, "slice_between": between, "alpha_from": a_from, "alpha_to": a_to, "_persistent": True}
        half = torch.full((B, 1), 0.5, device=self.device, dtype=torch.float32)
        return {"image": image, "slice_between": between, "alpha_from": half, "alpha_to": half.clone()}

    def next_batch(self, B, step=2, reuse_output=False, shard=None):
        """A random training batch: B random (volume, slice) pairs, neighbours ``step`` apart.  ``reuse_output``: the batch is written
        into this augmenter's persistent output buffer (valid until the next call) -- what the training loop uses.  ``shard`` = (rank,
        world): data parallel -- EVERY random number of the global batch is drawn (all ranks keep the same stream position and see the
        batch a single process would see), but only this rank's triplets [B r / W, B (r + 1) / W) are assembled: the host cost of a
        rank is the global batch's draws plus its own few descriptors, the device cost one launch over its own triplets."""
        trips = []
        for _ in range(B):
            vid = int(self.rs.randint(0, len(self.shapes)))
            Z = self.shapes[vid][0]
            if Z < step + 1:
                raise ValueError("volume %d has too few slices (%d) for step %d" % (vid, Z, step))
            sid = int(self.rs.randint(0, Z))
            zf, zt, zb = self.draw_triplet(vid, sid, step)
            trips.append((vid, zf, zt, zb))
        if shard is None:
            return self.assemble(trips, reuse_output=reuse_output)
        rank, world = int(shard[0]), int(shard[1])
        transforms = [self.draw_transform(t[0]) for t in trips]         # the order assemble() draws them in
        lo, hi = (B * rank) // world, (B * (rank + 1)) // world
        return self.assemble(trips[lo:hi], transforms[lo:hi], reuse_output=reuse_output)
