"""not-gpu: the SimpleITK-free NIfTI-1 / MetaImage reader-writer behind the inference CLI (volume_io.py).  Files are built
byte by byte from the published format layouts (NIfTI-1 header offsets; MetaImage key = value header) so the reader is not
checked against its own writer only."""
import gzip
import struct
import zlib

import numpy as np
import pytest

from superresolution_aniso_mri_amd import volume_io


def make_nifti(arr_zyx, pixdim, endian="<", sform=True, slope=0.0, inter=0.0, datatype=None):
    code = {np.dtype(np.int16): 4, np.dtype(np.float32): 16, np.dtype(np.uint8): 2, np.dtype(np.float64): 64}[arr_zyx.dtype] \
        if datatype is None else datatype
    hdr = bytearray(352)
    struct.pack_into(endian + "i", hdr, 0, 348)
    nd = arr_zyx.ndim
    struct.pack_into(endian + "8h", hdr, 40, nd, *(list(arr_zyx.shape[::-1]) + [1] * (7 - nd)))
    struct.pack_into(endian + "h", hdr, 70, code)
    struct.pack_into(endian + "h", hdr, 72, arr_zyx.dtype.itemsize * 8)
    struct.pack_into(endian + "8f", hdr, 76, 1.0, *(list(pixdim) + [0.0] * (7 - len(pixdim))))
    struct.pack_into(endian + "f", hdr, 108, 352.0)
    struct.pack_into(endian + "2f", hdr, 112, slope, inter)
    if sform:
        struct.pack_into(endian + "h", hdr, 254, 1)
        struct.pack_into(endian + "4f", hdr, 280, -pixdim[0], 0.0, 0.0, 10.0)
        struct.pack_into(endian + "4f", hdr, 296, 0.0, pixdim[1], 0.0, -20.0)
        struct.pack_into(endian + "4f", hdr, 312, 0.0, 0.0, pixdim[2], 5.0)
    hdr[344:348] = b"n+1\0"
    return bytes(hdr) + arr_zyx.astype(arr_zyx.dtype.newbyteorder(endian)).tobytes()


@pytest.mark.parametrize("endian", ["<", ">"])
@pytest.mark.parametrize("gz", [False, True])
def test_nifti_read_orientation_and_writeback(tmp_path, endian, gz):
    rng = np.random.default_rng(3)
    arr = rng.integers(-500, 3000, size=(5, 7, 9)).astype(np.int16)          # [z,y,x]
    blob = make_nifti(arr, (1.25, 1.5, 8.0), endian)
    p = tmp_path / ("v.nii.gz" if gz else "v.nii")
    (gzip.open if gz else open)(str(p), "wb").write(blob)
    vol = volume_io.read_volume(p)
    assert vol.array.shape == (5, 7, 9) and vol.array.dtype == np.int16
    assert np.array_equal(vol.array, arr)                                      # x is the fastest axis in the file
    assert vol.spacing == (1.25, 1.5, 8.0) and vol.num_frames == 1
    hr = rng.random((29, 7, 9)).astype(np.float32)                            # (5-1)*7+1 slices
    out = tmp_path / ("hr.nii.gz" if gz else "hr.nii")
    volume_io.write_volume(out, vol, hr, (1.25, 1.5, 8.0 / 7))
    raw = (gzip.open if gz else open)(str(out), "rb").read()
    assert struct.unpack(endian + "8h", raw[40:56])[:4] == (3, 9, 7, 29)
    assert struct.unpack(endian + "h", raw[70:72])[0] == 16 and struct.unpack(endian + "h", raw[72:74])[0] == 32
    assert abs(struct.unpack(endian + "8f", raw[76:108])[3] - 8.0 / 7) < 1e-6
    assert abs(struct.unpack(endian + "4f", raw[312:328])[2] - 8.0 / 7) < 1e-6    # sform z column follows the spacing
    assert struct.unpack(endian + "4f", raw[312:328])[3] == 5.0                   # origin untouched
    back = volume_io.read_volume(out)
    assert np.array_equal(back.array, hr) and abs(back.spacing[2] - 8.0 / 7) < 1e-6


def test_nifti_scaling_4d_and_errors(tmp_path):
    arr = np.arange(2 * 3 * 4 * 5, dtype=np.int16).reshape(2, 3, 4, 5)       # [t,z,y,x]
    p = tmp_path / "t.nii"
    p.write_bytes(make_nifti(arr, (1.0, 1.0, 5.0, 1.0), slope=0.5, inter=2.0))
    vol = volume_io.read_volume(p)
    assert vol.num_frames == 2 and vol.array.dtype == np.float32
    assert np.allclose(vol.array, arr * 0.5 + 2.0)
    (tmp_path / "bad.nii").write_bytes(b"\0" * 400)
    with pytest.raises(ValueError, match="not a NIfTI-1"):
        volume_io.read_volume(tmp_path / "bad.nii")
    with pytest.raises(ValueError, match="unsupported volume file"):
        volume_io.read_volume(tmp_path / "x.dcm")


@pytest.mark.parametrize("compressed", [False, True])
def test_metaimage_mha_and_mhd(tmp_path, compressed):
    rng = np.random.default_rng(4)
    arr = rng.random((4, 6, 8)).astype(np.float32)
    payload = zlib.compress(arr.tobytes()) if compressed else arr.tobytes()
    head = ("ObjectType = Image\nNDims = 3\nBinaryData = True\nBinaryDataByteOrderMSB = False\nCompressedData = %s\n"
            "TransformMatrix = 1 0 0 0 1 0 0 0 1\nOffset = -3 4 5\nElementSpacing = 0.9 0.9 6\nDimSize = 8 6 4\n"
            "ElementType = MET_FLOAT\nElementDataFile = LOCAL\n" % ("True" if compressed else "False"))
    p = tmp_path / "v.mha"
    p.write_bytes(head.encode() + payload)
    vol = volume_io.read_volume(p)
    assert np.array_equal(vol.array, arr) and vol.spacing == (0.9, 0.9, 6.0)
    hr = rng.random((10, 6, 8)).astype(np.float32)
    out = tmp_path / "hr.mhd"
    volume_io.write_volume(out, vol, hr, (0.9, 0.9, 2.0))
    text = out.read_text()
    assert "DimSize = 8 6 10" in text and "ElementDataFile = hr.raw" in text and "Offset = -3 4 5" in text
    assert "CompressedData = False" in text and "ElementSpacing = 0.9 0.9 2.0" in text
    back = volume_io.read_volume(out)
    assert np.array_equal(back.array, hr) and back.spacing == (0.9, 0.9, 2.0)
    out2 = tmp_path / "hr.mha"
    volume_io.write_volume(out2, vol, hr, (0.9, 0.9, 2.0))
    assert np.array_equal(volume_io.read_volume(out2).array, hr)
