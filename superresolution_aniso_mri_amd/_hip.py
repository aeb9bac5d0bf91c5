"""ctypes binding of libaesr_hip.so (C ABI: include/aesr_hip.h).

There is NO fallback: if the shared library is missing or an entry point is absent this module raises at
import time, and every wrapper raises RuntimeError with the library's own message when a call fails.
PyTorch is used only for device memory and the current HIP stream.
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libaesr_hip.so")
if os.environ.get("AESR_LIB"):
    # experiments only (scripts/variants.py): a variant build of the library is loaded INSTEAD of the shipped one -- the shipped file
    # is never overwritten by an A/B script -- and every process that does so says it on stderr
    import sys as _sys
    LIB_PATH = os.path.abspath(os.environ["AESR_LIB"])
    _sys.stderr.write("[aesr] AESR_LIB: loading the VARIANT library %s (not the shipped libaesr_hip.so)\n" % LIB_PATH)

P = c_void_p          # device pointer
IP = ctypes.POINTER(c_int)
FP = ctypes.POINTER(c_float)
DP = ctypes.POINTER(c_double)     # host array of doubles

class TripletDesc(ctypes.Structure):
    """aesr_triplet_desc of include/aesr_hip.h"""
    _fields_ = [("vol_off", ctypes.c_longlong), ("H", c_int), ("W", c_int), ("z_from", c_int), ("z_to", c_int), ("z_between", c_int),
                ("oy", c_int), ("ox", c_int), ("k", c_int), ("gain", c_float), ("cutoff", c_float)]


class WgradReduceJob(ctypes.Structure):
    """aesr_wgrad_reduce_job of include/aesr_hip.h"""
    _fields_ = [("workspace", c_void_p), ("dw", c_void_p), ("db", c_void_p), ("N", c_int), ("H", c_int), ("W", c_int), ("Cin", c_int),
                ("Cout", c_int), ("KS", c_int), ("pad", c_int)]


class PrepJob(ctypes.Structure):
    """aesr_prep_job of include/aesr_hip.h"""
    _fields_ = [("w", c_void_p), ("aux0", c_void_p), ("aux1", c_void_p), ("out", c_void_p), ("kind", c_int), ("Cout", c_int), ("Cin", c_int),
                ("KS", c_int), ("transpose", c_int)]


PREP_PACK, PREP_WINO_PACK, PREP_STEM_FOLD, PREP_COUT1_FLIP = 0, 1, 2, 3


class PackJob(ctypes.Structure):
    """aesr_pack_job of include/aesr_hip.h"""
    _fields_ = [("w", c_void_p), ("packed", c_void_p), ("Cout", c_int), ("Cin", c_int), ("KS", c_int), ("transpose", c_int)]


# name -> (restype, argtypes); must list every symbol include/aesr_hip.h declares (checked by tests)
SIGNATURES = {
    "aesr_version": (c_int, []),
    "aesr_last_error_string": (c_char_p, []),
    "aesr_conv2d_packed_floats": (c_size_t, [c_int, c_int, c_int, c_int]),
    "aesr_conv2d_pack": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "aesr_conv2d_pack_many": (c_int, [ctypes.POINTER(PackJob), c_int, P]),
    "aesr_weight_prep_many": (c_int, [ctypes.POINTER(PrepJob), c_int, P]),
    "aesr_conv2d_cout1_dgrad_pre": (c_int, [P, P, P, P] + [c_int] * 5 + [c_float, P]),
    "aesr_conv2d_fwd": (c_int, [P, P, P, P] + [c_int] * 8 + [c_float, P]),
    "aesr_conv2d_workspace_floats": (c_size_t, [c_int] * 7),
    "aesr_conv2d_dgrad_workspace_floats": (c_size_t, [c_int] * 7),
    "aesr_conv2d_fwd_ws": (c_int, [P, P, P, P, P] + [c_int] * 8 + [c_float, P]),
    "aesr_conv2d_dgrad_ws": (c_int, [P, P, P, P, P] + [c_int] * 8 + [c_float, P]),
    "aesr_conv2d_dgrad": (c_int, [P, P, P, P] + [c_int] * 8 + [c_float, P]),
    "aesr_conv2d_wgrad_workspace_floats": (c_size_t, [c_int] * 7),
    "aesr_conv2d_wgrad": (c_int, [P, P, P, P, P] + [c_int] * 7 + [P]),
    "aesr_conv2d_smallcin_fwd": (c_int, [P, P, P, P, P] + [c_int] * 9 + [c_float, c_int, c_int, FP, FP, P]),
    "aesr_conv2d_smallcin_dgrad": (c_int, [P, P, P] + [c_int] * 8 + [FP, P]),
    "aesr_small_wgrad_workspace_floats": (c_size_t, [c_int]),
    "aesr_conv2d_smallcin_wgrad": (c_int, [P, P, P, P, P] + [c_int] * 6 + [P]),
    "aesr_conv2d_cout1_fwd": (c_int, [P, P, P, P] + [c_int] * 5 + [c_float, P]),
    "aesr_conv2d_cout1_workspace_floats": (c_size_t, [c_int]),
    "aesr_conv2d_cout1_wgrad": (c_int, [P, P, P, P, P] + [c_int] * 4 + [P]),
    "aesr_conv2d_cout1_dgrad": (c_int, [P, P, P, P, P] + [c_int] * 5 + [c_float, P]),
    "aesr_stemconv_folded_floats": (c_size_t, [c_int]),
    "aesr_stemconv_fold": (c_int, [P, P, P, P, c_int, c_int, P]),
    "aesr_stemconv_fwd": (c_int, [P, P, P, P] + [c_int] * 6 + [c_float, P]),
    "aesr_stemconv_workspace_floats": (c_size_t, [c_int]),
    "aesr_stemconv_wgrad": (c_int, [P] * 10 + [c_int] * 6 + [P]),
    "aesr_space_to_depth2": (c_int, [P, P] + [c_int] * 4 + [P]),
    "aesr_depth_to_space2": (c_int, [P, P] + [c_int] * 4 + [P]),
    "aesr_resample2_fwd": (c_int, [P, P] + [c_int] * 5 + [P]),
    "aesr_resample2_bwd": (c_int, [P, P, P] + [c_int] * 6 + [c_float, P]),
    "aesr_bn_stats": (c_int, [P, P, P, c_int, c_int, c_int, IP, P]),
    "aesr_bn_finalize": (c_int, [P, DP] + [P] * 9 + [c_int, c_int, c_float, c_float, c_int, c_int, P]),
    "aesr_bn_stats_finalize": (c_int, [P, P, DP] + [P] * 9 + [c_int, c_int, c_int, IP, c_float, c_float, c_int, P]),
    "aesr_bn_bwd": (c_int, [P] * 6 + [DP] + [P] * 4 + [c_int] * 6 + [c_float, c_int, IP, P]),
    "aesr_bn_fused_supported": (c_int, [c_int, c_int]),
    "aesr_bn_finalize_apply": (c_int, [P] * 13 + [c_int] * 6 + [P, c_float, c_float, c_int, P]),
    "aesr_bn_fused1_supported": (c_int, [c_int] * 7),
    "aesr_bn_fused1_workspace_floats": (c_size_t, [c_int, c_int]),
    "aesr_bn_fused1_barrier_words": (c_size_t, []),
    "aesr_bn_fused1_timeouts": (ctypes.c_uint, []),
    "aesr_bn_fused1_fwd": (c_int, [P, P, P, P, DP] + [P] * 9 + [c_int] * 6 + [IP, c_float, c_float, c_int, P]),
    "aesr_bn_fused1_bwd": (c_int, [P] * 7 + [DP] + [P] * 4 + [c_int] * 6 + [c_float, c_int, IP, P]),
    "aesr_p2p_alloc": (c_int, [c_size_t, ctypes.POINTER(c_void_p)]),
    "aesr_p2p_free": (c_int, [P]),
    "aesr_p2p_get_handle": (c_int, [P, c_char_p]),
    "aesr_p2p_open": (c_int, [c_char_p, ctypes.POINTER(c_void_p)]),
    "aesr_p2p_close": (c_int, [P]),
    "aesr_p2p_region_bytes": (c_size_t, [c_int]),
    "aesr_p2p_tick": (c_int, [P, P]),
    "aesr_bn_fused1_fwd_p2p": (c_int, [P, P, P, P, DP] + [P] * 9 + [c_int] * 6 + [IP, c_float, c_float, c_int, ctypes.POINTER(c_void_p), c_int, c_int, c_int, P, P]),
    "aesr_bn_fused1_bwd_p2p": (c_int, [P] * 7 + [DP] + [P] * 4 + [c_int] * 6 + [c_float, c_int, IP, ctypes.POINTER(c_void_p), c_int, c_int, c_int, P, P]),
    "aesr_bn_apply": (c_int, [P, P, P, P] + [c_int] * 6 + [IP, P]),
    "aesr_bn_bwd_reduce": (c_int, [P] * 6 + [c_int] * 6 + [IP, P]),
    "aesr_bn_bwd_apply": (c_int, [P] * 6 + [DP] + [P] * 4 + [c_int] * 6 + [c_float, c_int, IP, P]),
    "aesr_scale_expand_fwd": (c_int, [P, P, c_size_t, FP, FP, P]),
    "aesr_scale_expand_bwd": (c_int, [P, P, c_size_t, FP, P]),
    "aesr_maxpool2_fwd": (c_int, [P, P] + [c_int] * 4 + [P]),
    "aesr_maxpool2_bwd": (c_int, [P, P, P, P] + [c_int] * 5 + [P]),
    "aesr_lpips_tap_fwd": (c_int, [P, P, P, c_int, c_int, c_int, P]),
    "aesr_lpips_tap_bwd": (c_int, [P, P, P, P, c_int, c_int, c_int, P]),
    "aesr_lpips_finalize": (c_int, [ctypes.POINTER(c_void_p), IP, c_int, P, c_int, P]),
    "aesr_lerp_fwd": (c_int, [P, P, P, P, c_int, c_size_t, P]),
    "aesr_lerp_bwd": (c_int, [P, P, P, P, c_int, c_size_t, P]),
    "aesr_lerp_cat_fwd": (c_int, [P, P, P, P, c_int, c_size_t, P]),
    "aesr_lerp_multi": (c_int, [P, P, c_int, c_size_t, FP, c_int, c_int, c_float, P]),
    "aesr_lerp_cat_bwd": (c_int, [P, P, P, P, c_int, c_size_t, P]),
    "aesr_interleave_clamp": (c_int, [P, P, P, c_int, c_int, c_size_t, c_float, c_float, P]),
    "aesr_mse_fwd": (c_int, [P, P, P, P, c_size_t, P]),
    "aesr_mse_bwd": (c_int, [P, P, P, P, c_size_t, P]),
    "aesr_mse3_fwd": (c_int, [P, P, c_size_t, P, P, c_size_t, P, P, c_size_t, P, P, P, P]),
    "aesr_mse3_bwd": (c_int, [P, P, c_size_t, P, P, c_size_t, P, P, P, P, P]),
    "aesr_row_mean_fwd": (c_int, [P, P, c_int, c_size_t, P]),
    "aesr_row_mean_bwd": (c_int, [P, P, c_int, c_size_t, P]),
    "aesr_l1_fwd": (c_int, [P, P, P, P, c_size_t, P]),
    "aesr_l1_bwd": (c_int, [P, P, P, P, c_size_t, P]),
    "aesr_lap_blur5": (c_int, [P, P, P, c_int, c_int, c_int, c_float, c_int, P]),
    "aesr_lap_down2": (c_int, [P, P, c_int, c_int, c_int, P]),
    "aesr_lap_zero_insert2": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "aesr_act_bwd": (c_int, [P, P, P, c_size_t, c_int, c_float, P]),
    "aesr_triplet_assemble": (c_int, [P, ctypes.POINTER(TripletDesc), c_int, c_int, P, P, P]),
    "aesr_ssim_workspace_doubles": (c_size_t, [c_int, c_int, c_int]),
    "aesr_ssim_mse": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_double, c_double, c_double, P]),
    "aesr_vif_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "aesr_vif_mscale": (c_int, [P, P, P, P, c_int, c_int, c_int, DP, IP, c_double, P]),
    "aesr_adam_state_init": (None, [FP, c_double, c_double, c_double]),
    "aesr_adam_step": (c_int, [P, P, P, P, P, c_size_t, c_float, c_double, c_double, c_float, c_float, c_int, P]),
    "aesr_conv2d_wino_supported": (c_int, [c_int] * 5),
    "aesr_conv2d_wino_kernel": (c_int, [c_int] * 8),
    "aesr_conv2d_wino_ring_timeouts": (ctypes.c_uint, []),
    "aesr_conv2d_wino_packed_floats": (c_size_t, [c_int, c_int, c_int]),
    "aesr_conv2d_wino_pack_many": (c_int, [ctypes.POINTER(PackJob), c_int, P]),
    "aesr_conv2d_wino_fwd": (c_int, [P, P, P, P] + [c_int] * 6 + [c_float, P]),
    "aesr_conv2d_wino_dgrad": (c_int, [P, P, P, P] + [c_int] * 6 + [c_float, P]),
    "aesr_conv2d_wino_workspace_floats": (c_size_t, [c_int] * 6),
    "aesr_conv2d_wino_fwd_bn_supported": (c_int, [c_int] * 5),
    "aesr_conv2d_wino_fwd_bn": (c_int, [P, P, P, P, P, P] + [c_int] * 6 + [c_float, c_int, P]),
    "aesr_conv2d_wino_fwd_ws": (c_int, [P, P, P, P, P, c_size_t] + [c_int] * 6 + [c_float, P]),
    "aesr_conv2d_wino_dgrad_ws": (c_int, [P, P, P, P, P, c_size_t] + [c_int] * 6 + [c_float, P]),
    "aesr_conv2d_wino_fwd_up2": (c_int, [P, P, P, P] + [c_int] * 6 + [c_float, P]),
    "aesr_conv2d_wino_dgrad_sum2": (c_int, [P, P, P] + [c_int] * 5 + [P]),
    "aesr_conv2d_wgrad_up2_supported": (c_int, [c_int, c_int]),
    "aesr_conv2d_wgrad_up2": (c_int, [P, P, P, P, P] + [c_int] * 5 + [P]),
    "aesr_conv2d_wgrad_partial": (c_int, [P, P, P] + [c_int] * 8 + [P]),
    "aesr_conv2d_wgrad_reduce_many": (c_int, [ctypes.POINTER(WgradReduceJob), c_int, P]),
    "aesr_comm_rccl_version": (c_int, [IP]),
    "aesr_comm_unique_id": (c_int, [c_char_p]),
    "aesr_comm_init": (c_int, [c_char_p, c_int, c_int, ctypes.POINTER(c_void_p)]),
    "aesr_comm_destroy": (c_int, [P]),
    "aesr_comm_abort": (c_int, [P]),
    "aesr_comm_allreduce": (c_int, [P, P, c_size_t, c_int, c_int, P]),
    "aesr_comm_allreduce_many": (c_int, [P, ctypes.POINTER(c_void_p), ctypes.POINTER(c_size_t), c_int, c_int, c_int, P]),
    "aesr_comm_broadcast": (c_int, [P, P, c_size_t, c_int, c_int, P]),
}
P2P_HANDLE_BYTES, P2P_SLOTS = 64, 32
COMM_ID_BYTES = 128
COMM_F32, COMM_F64, COMM_SUM, COMM_MAX = 0, 1, 0, 1

ACT_NONE, ACT_LRELU, ACT_RELU, ACT_SIGMOID = 0, 1, 2, 3
BN_NONE, BN_POOL, BN_UP = 0, 1, 2
RS_POOL, RS_NEAREST, RS_BILINEAR = 1, 2, 3
BN_NWG = 512
REDUCE_MAX_JOBS = 16          # REDUCE_MAX_JOBS of csrc/aesr_kernels.h: layers per aesr_conv2d_wgrad_reduce_many launch
MSE_NPART = 512
MSE3_WS = 3 * 256 + 1            # doubles (AESR_MSE3_WS)
LPIPS_NCH = 64


def _load():
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libaesr_hip.so not found at %s -- build it with `python __graft_entry__.py` (or `make -C "
            "superresolution_aniso_mri_amd/csrc`). The HIP extension is mandatory: there is no CPU/eager fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            raise RuntimeError("libaesr_hip.so does not export %s (stale build?)" % name)
        fn.restype, fn.argtypes = res, args
    return lib


lib = _load()


def last_error():
    return lib.aesr_last_error_string().decode("utf-8", "replace")


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed (code %d): %s" % (what, rc, last_error()))


def check_device_watchdogs(where):
    """Raise if a kernel-side protocol watchdog has fired in this process.  The ring kernel's waves wait on LDS arrival counters
    (csrc/conv_wino_ring.hip); a wait that gives up -- a protocol bug, never seen -- lets the wave go on with an incomplete filter
    chunk rather than hang the GPU, and counts.  Every place where results LEAVE the process (checkpoints, logged epoch means,
    validation, synthesised volumes, the bench line) calls this, so garbage cannot be trained on or written silently.  Reads a
    device symbol (synchronises the device): never inside the step or a graph capture."""
    nb = int(lib.aesr_bn_fused1_timeouts())
    if nb:
        raise RuntimeError("%s: the one-launch BatchNorm kernel's grid barrier gave up %d time(s) in this process -- its workgroups were not all "
                           "resident (the device is shared with another grid-barrier kernel: set AESR_BN_FUSED=0); results since are not "
                           "trustworthy; nothing was written" % (where, nb))
    n = int(lib.aesr_conv2d_wino_ring_timeouts())
    if n:
        raise RuntimeError("%s: the ring convolution kernel's arrival-counter watchdog fired %d time(s) in this process -- activations / "
                           "gradients computed since are not trustworthy (csrc/conv_wino_ring.hip: wait_full); nothing was written" % (where, n))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream():
    """The current HIP stream of the current device as a void*.  ``torch.cuda.current_stream()`` builds a Stream object per call
    (~9 us, several hundred calls per host-launched step); the raw accessor behind it takes ~0.5 us."""
    if _raw_stream is not None and _cur_device is not None:
        return c_void_p(_raw_stream(_cur_device()))
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def int_array(vals):
    return (c_int * len(vals))(*[int(v) for v in vals])


def double_array(vals):
    return (c_double * len(vals))(*[float(v) for v in vals])


def float_array(vals):
    return (c_float * len(vals))(*[float(v) for v in vals])


def require_gpu_tensor(t, name, dtype=torch.float32):
    if not t.is_cuda:
        raise RuntimeError("%s must live on the GPU (got %s): the HIP path has no CPU fallback" % (name, t.device))
    if t.dtype != dtype:
        raise RuntimeError("%s must be %s (got %s)" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)
    return t
