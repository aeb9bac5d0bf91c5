"""Minimal NIfTI-1 (.nii / .nii.gz) and MetaImage (.mha / .mhd+raw) reader-writer in numpy, so that the inference CLI
(generate_hr_volumes.py:104-183 of the reference: ``sitk.ReadImage`` / ``numpy_to_sitk`` / ``sitk.WriteImage``) runs where
SimpleITK is not installed.  Only what that path needs: scalar 3-D / 4-D volumes, the voxel array in SimpleITK's
``GetArrayFromImage`` axis order ([z,y,x] or [t,z,y,x]), spacing / origin / direction carried through, and writing the
through-plane up-sampled volume with the z spacing divided by (n+1) (generate_hr_volumes.py:177-181).

``Volume.array`` is always [z,y,x] or [t,z,y,x]; ``Volume.spacing`` is (x, y, z[, t]) like ``sitk.Image.GetSpacing()``.
When SimpleITK is importable the CLI keeps using it (identical files to the reference); this module is the fallback."""
import gzip
import os
import struct
import zlib

import numpy as np

_NIFTI_DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16,
                 768: np.uint32, 1024: np.int64, 1280: np.uint64}
_NIFTI_CODES = {np.dtype(v).str[1:]: k for k, v in _NIFTI_DTYPES.items()}
_MET_TYPES = {"MET_UCHAR": np.uint8, "MET_CHAR": np.int8, "MET_USHORT": np.uint16, "MET_SHORT": np.int16, "MET_UINT": np.uint32,
              "MET_INT": np.int32, "MET_ULONG": np.uint64, "MET_LONG": np.int64, "MET_FLOAT": np.float32, "MET_DOUBLE": np.float64}
_MET_NAMES = {np.dtype(v).str[1:]: k for k, v in _MET_TYPES.items()}


class Volume:
    """array [z,y,x] or [t,z,y,x]; spacing (x,y,z[,t]); ``meta`` = format-specific header state used when writing back."""

    def __init__(self, array, spacing, fmt, meta):
        self.array, self.spacing, self.fmt, self.meta = array, tuple(float(s) for s in spacing), fmt, meta

    @property
    def num_frames(self):
        return 1 if self.array.ndim == 3 else self.array.shape[0]


# ---- NIfTI-1 ----------------------------------------------------------------------------------------------------------
def _read_nifti(path):
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(str(path), "rb") as f:
        raw = f.read()
    if len(raw) < 352:
        raise ValueError("%s: too short for a NIfTI-1 header" % path)
    end = "<"
    if struct.unpack("<i", raw[:4])[0] != 348:
        end = ">"
        if struct.unpack(">i", raw[:4])[0] != 348:
            raise ValueError("%s: not a NIfTI-1 file (sizeof_hdr != 348)" % path)
    if raw[344:348] not in (b"n+1\0",):
        raise ValueError("%s: only single-file NIfTI-1 (magic n+1) is supported" % path)
    dim = struct.unpack(end + "8h", raw[40:56])
    datatype = struct.unpack(end + "h", raw[70:72])[0]
    pixdim = struct.unpack(end + "8f", raw[76:108])
    vox_offset = int(struct.unpack(end + "f", raw[108:112])[0])
    slope, inter = struct.unpack(end + "2f", raw[112:120])
    nd = dim[0]
    if nd not in (3, 4) or datatype not in _NIFTI_DTYPES:
        raise ValueError("%s: need a scalar 3-D or 4-D volume (dim[0]=%d, datatype=%d)" % (path, nd, datatype))
    shape = dim[1:1 + nd]
    dt = np.dtype(_NIFTI_DTYPES[datatype]).newbyteorder(end)
    n = int(np.prod(shape))
    data = np.frombuffer(raw, dtype=dt, count=n, offset=vox_offset).reshape(shape[::-1])     # x fastest -> [t,]z,y,x
    data = data.astype(dt.newbyteorder("="))
    if slope not in (0.0, 1.0) or inter != 0.0:
        if slope != 0.0:
            data = data.astype(np.float32) * np.float32(slope) + np.float32(inter)
    return Volume(data, pixdim[1:1 + nd], "nifti", {"header": bytes(raw[:vox_offset]), "endian": end})


def _write_nifti(path, vol, array, spacing):
    end = vol.meta["endian"] if vol.fmt == "nifti" else "<"
    hdr = bytearray(vol.meta["header"]) if vol.fmt == "nifti" else bytearray(352)
    if vol.fmt != "nifti":
        struct.pack_into("<i", hdr, 0, 348)
        hdr[344:348] = b"n+1\0"
        struct.pack_into("<f", hdr, 108, 352.0)
        struct.pack_into("<f", hdr, 76, 1.0)                       # qfac
        struct.pack_into("<h", hdr, 252, 0)                        # qform_code 0, sform_code set below
    key = np.dtype(array.dtype).str[1:]
    if key not in _NIFTI_CODES:
        raise ValueError("no NIfTI datatype for %s" % array.dtype)
    nd = array.ndim
    dims = [nd] + list(array.shape[::-1]) + [1] * (7 - nd)
    struct.pack_into(end + "8h", hdr, 40, *dims)
    struct.pack_into(end + "h", hdr, 70, _NIFTI_CODES[key])
    struct.pack_into(end + "h", hdr, 72, array.dtype.itemsize * 8)
    old_pix = struct.unpack(end + "8f", bytes(hdr[76:108]))
    pix = list(old_pix)
    for i, s in enumerate(spacing):
        pix[1 + i] = float(s)
    struct.pack_into(end + "8f", hdr, 76, *pix)
    struct.pack_into(end + "2f", hdr, 112, 1.0, 0.0)               # data are stored unscaled
    # sform rows (srow_x/y/z at 280/296/312): the z column carries direction * spacing_z -> rescale it with the spacing
    sform_code = struct.unpack(end + "h", bytes(hdr[254:256]))[0]
    if sform_code > 0 and old_pix[3] > 0 and len(spacing) >= 3:
        ratio = float(spacing[2]) / old_pix[3]
        for off in (280, 296, 312):
            row = list(struct.unpack(end + "4f", bytes(hdr[off:off + 16])))
            row[2] *= ratio
            struct.pack_into(end + "4f", hdr, off, *row)
    vox_offset = int(struct.unpack(end + "f", bytes(hdr[108:112]))[0])
    hdr = hdr[:vox_offset] if len(hdr) >= vox_offset else hdr + bytearray(vox_offset - len(hdr))
    payload = bytes(hdr) + np.ascontiguousarray(array.astype(array.dtype.newbyteorder(end))).tobytes()
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(str(path), "wb") as f:
        f.write(payload)


# ---- MetaImage --------------------------------------------------------------------------------------------------------
def _read_meta(path):
    with open(str(path), "rb") as f:
        raw = f.read()
    fields, pos = {}, 0
    order = []
    while True:
        nl = raw.index(b"\n", pos)
        line = raw[pos:nl].decode("ascii", "replace").strip()
        pos = nl + 1
        if "=" not in line:
            continue
        k, v = [t.strip() for t in line.split("=", 1)]
        fields[k] = v
        order.append(k)
        if k == "ElementDataFile":
            break
    nd = int(fields["NDims"])
    if nd not in (3, 4) or fields.get("ElementType") not in _MET_TYPES or int(fields.get("ElementNumberOfChannels", "1")) != 1:
        raise ValueError("%s: need a scalar 3-D or 4-D MetaImage" % path)
    shape = [int(t) for t in fields["DimSize"].split()]
    spacing = [float(t) for t in fields.get("ElementSpacing", fields.get("ElementSize", " ".join(["1"] * nd))).split()]
    msb = fields.get("BinaryDataByteOrderMSB", fields.get("ElementByteOrderMSB", "False")).lower() == "true"
    dt = np.dtype(_MET_TYPES[fields["ElementType"]]).newbyteorder(">" if msb else "<")
    if fields["ElementDataFile"] == "LOCAL":
        blob = raw[pos:]
    else:
        with open(os.path.join(os.path.dirname(str(path)), fields["ElementDataFile"]), "rb") as f:
            blob = f.read()
    if fields.get("CompressedData", "False").lower() == "true":
        blob = zlib.decompress(blob)
    n = int(np.prod(shape))
    data = np.frombuffer(blob, dtype=dt, count=n).reshape(shape[::-1]).astype(dt.newbyteorder("="))
    return Volume(data, spacing, "meta", {"fields": fields, "order": order})


def _write_meta(path, vol, array, spacing):
    key = np.dtype(array.dtype).str[1:]
    if key not in _MET_NAMES:
        raise ValueError("no MetaImage element type for %s" % array.dtype)
    nd = array.ndim
    fields = dict(vol.meta["fields"]) if vol.fmt == "meta" else {"ObjectType": "Image", "BinaryData": "True"}
    order = list(vol.meta["order"]) if vol.fmt == "meta" else ["ObjectType", "NDims", "BinaryData", "BinaryDataByteOrderMSB",
                                                                "CompressedData", "ElementSpacing", "DimSize", "ElementType",
                                                                "ElementDataFile"]
    fields.update({"NDims": str(nd), "DimSize": " ".join(str(v) for v in array.shape[::-1]),
                   "ElementSpacing": " ".join(repr(float(s)) for s in spacing), "ElementType": _MET_NAMES[key],
                   "BinaryDataByteOrderMSB": "False", "CompressedData": "False"})
    fields.pop("CompressedDataSize", None)
    fields.pop("ElementByteOrderMSB", None)
    order = [k for k in order if k in fields and k != "ElementDataFile"]
    for k in ("NDims", "BinaryDataByteOrderMSB", "CompressedData", "ElementSpacing", "DimSize", "ElementType"):
        if k not in order:
            order.append(k)
    blob = np.ascontiguousarray(array.astype(array.dtype.newbyteorder("<"))).tobytes()
    path = str(path)
    if path.endswith(".mhd"):
        rawname = os.path.basename(path)[:-4] + ".raw"
        fields["ElementDataFile"] = rawname
        with open(os.path.join(os.path.dirname(path), rawname), "wb") as f:
            f.write(blob)
        blob = b""
    else:
        fields["ElementDataFile"] = "LOCAL"
    head = "".join("%s = %s\n" % (k, fields[k]) for k in order + ["ElementDataFile"])
    with open(path, "wb") as f:
        f.write(head.encode("ascii") + blob)


# ---- public -----------------------------------------------------------------------------------------------------------
def read_volume(path):
    p = str(path).lower()
    if p.endswith(".nii") or p.endswith(".nii.gz"):
        return _read_nifti(path)
    if p.endswith(".mha") or p.endswith(".mhd"):
        return _read_meta(path)
    raise ValueError("unsupported volume file %s (nii, nii.gz, mha, mhd)" % path)


def write_volume(path, like, array, spacing=None):
    """Write ``array`` ([z,y,x] or [t,z,y,x]) with the geometry of ``like`` (a Volume) and, if given, a new spacing."""
    spacing = like.spacing[:array.ndim] if spacing is None else tuple(spacing)
    p = str(path).lower()
    if p.endswith(".nii") or p.endswith(".nii.gz"):
        return _write_nifti(path, like, np.asarray(array), spacing)
    if p.endswith(".mha") or p.endswith(".mhd"):
        return _write_meta(path, like, np.asarray(array), spacing)
    raise ValueError("unsupported volume file %s (nii, nii.gz, mha, mhd)" % path)
