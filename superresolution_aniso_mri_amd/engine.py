"""Host-side executor: runs a reference-shaped ``nn.Sequential`` (conv / LeakyReLU / BatchNorm / AvgPool /
Upsample / Sigmoid stacks of networks/acai_vanilla.py:49-102) on the HIP kernels of libaesr_hip.so.

The ``nn.Sequential`` only *holds* the parameters (so state_dict keys, parameter order and the reference's
init are untouched); none of its modules' ``forward`` is ever called.  The stack is compiled
into fused steps

    conv (+LeakyReLU | ReLU | Sigmoid epilogue)        -> aesr_conv2d_fwd / aesr_conv2d_smallcin_fwd
    BatchNorm (+AvgPool2d(2) | nearest Upsample x2)    -> aesr_bn_stats / finalize / apply

and a hand-written backward walks them in reverse (dgrad with the previous activation's derivative fused,
MFMA split-K wgrad, BatchNorm backward fused with the LeakyReLU derivative).  Activations are NHWC fp32.
A pass may carry several *statistic groups* (sub-batches with independent BatchNorm batch statistics); only a
leading prefix of the batch needs gradients.
"""
import os
from ctypes import c_float

import torch
import torch.nn as nn

from . import _hip
from ._hip import lib, ptr, stream, check

LRELU_SLOPE = 0.01


class KernelProfiler:
    """Optional HIP-event timing of the step's launches (bench.py roofline): one event pair per call on the launching stream; durations are
    read after a synchronize.  MFMA convolution launches carry their algorithmic flops, the bandwidth-bound calls of the step's tail
    (BatchNorm, the single-channel-side "thin" convolutions, slab sums, losses, lerp, Adam, the LPIPS head) their ALGORITHMIC BYTES: every
    tensor of the call read or written once.  Off (None) in normal operation."""

    def __init__(self):
        self.records = []      # (kind, flops, bytes, start_event, end_event)

    def begin(self, kind, flops, nbytes=0.0):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream())
        self.records.append((kind, flops, nbytes, e0, e1))

    def end(self):
        self.records[-1][4].record(torch.cuda.current_stream())

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for kind, flops, nbytes, e0, e1 in self.records:
            d = out.setdefault(kind, {"launches": 0, "flops": 0.0, "bytes": 0.0, "ms": 0.0})
            d["launches"] += 1
            d["flops"] += flops
            d["bytes"] += nbytes
            d["ms"] += e0.elapsed_time(e1)
        return out


PROFILER = None
# diagnostics / parity tests: a list collects (nn.Sequential, conv module, activation code, NHWC output tensor) for every convolution a pass
# runs -- the values whose signs ARE the path's LeakyReLU decisions (the backward masks are taken from these very tensors)
TRACE = None
FUSE_STEM = os.environ.get("AESR_FUSE_STEM", "1") != "0"      # fold the encoder stem into the first 3x3 conv
# 3x3 / padding-1 convolutions with enough channels run in Winograd F(2x2,3x3) form (csrc/conv_wino.hip: 2.25x fewer MFMA flops,
# results equal to the implicit GEMM within 2-4e-7); AESR_WINO=0 keeps every layer on the exact-fp32 fma-chain implicit GEMM
USE_WINO = os.environ.get("AESR_WINO", "1") != "0"
FOLD_UPSAMPLE = os.environ.get("AESR_FOLD_UPSAMPLE", "1") != "0"
# inference: conv + activation + eval-mode BatchNorm (+ AvgPool2d(2)) as ONE launch where the resident-filter kernel serves the layer
FUSE_EVAL_BN = os.environ.get("AESR_FUSE_EVAL_BN", "1") != "0"


def bn_fused_enabled():
    """Small batches run a BatchNorm call as ONE launch per direction (csrc/bn_fused.hip: activations resident in LDS, one grid-wide
    barrier) instead of three.  The kernel needs all 256 workgroups resident at once: a process that steps SEVERAL trainers concurrently
    on different streams of one device must switch it off (AESR_BN_FUSED=0) -- two such grids could each hold half of the chip and wait
    for the other half until their bounded waits give up (counted; _hip.check_device_watchdogs raises).  Read per call."""
    return os.environ.get("AESR_BN_FUSED", "1") != "0"


def bn_fused_pays(N, H, W, C):
    """Single process: the one-launch form only where it is not a loss.  It removes two launches and the second read of the layer, but its
    256 workgroups x 512 threads stream a layer at 1.7-1.8 TB/s where the three launches reach 4-5: at 12 triplets it LOSES 15 us per C2
    step and 15-20 us per C3 step (profiles/r05_bn_fused_ab.txt: same box, three alternating rounds), at 6 triplets 9 us; at 3 triplets
    it wins 5 us, below that nothing either way (profiles/r05_bn_fused_threshold.txt).  Default: calls of at most 9 images
    (AESR_BN_FUSED_MAX_IMAGES; AESR_BN_FUSED_MAX_MB bounds the layer's bytes for experiments).  The opt-in peer exchange
    (AESR_SYNCBN=p2p) needs the kernel and keeps its own rule (_p2p_fits)."""
    if N > int(os.environ.get("AESR_BN_FUSED_MAX_IMAGES", "9")):
        return False
    return float(N) * H * W * C * 4.0 <= float(os.environ.get("AESR_BN_FUSED_MAX_MB", "1000")) * 1e6


def wino_ok(cin, cout, ks, pad, transpose):
    return bool(USE_WINO and lib.aesr_conv2d_wino_supported(int(cin), int(cout), int(ks), int(pad), int(transpose)))


def wino_kind(n, h, w, cin, cout, transpose):
    """Profiler label of the Winograd kernel the library runs for a layer of n images of h x w outputs (conv_wino.hip /
    conv_wino_res.hip / conv_wino_ring.hip)."""
    k = lib.aesr_conv2d_wino_kernel(int(n), int(h), int(w), int(cin), int(cout), 3, 1, int(transpose))
    return {2: "conv_wino_res_f32", 3: "conv_wino_ring_f32"}.get(k, "conv_wino_f32")


def wino_workspace(like, n, h, w, cin, cout, transpose):
    """(workspace tensor or None, its floats) for aesr_conv2d_wino_fwd_ws / _dgrad_ws: the slabs of the ring kernel's channel split,
    which the library asks for on small layers with many K-side channels (0 floats elsewhere)."""
    nws = lib.aesr_conv2d_wino_workspace_floats(n, h, w, cin, cout, transpose)
    return (torch.empty((nws,), device=like.device, dtype=torch.float32), nws) if nws else (None, 0)


def wgrad_kind(cin, cout, ks, pad):
    """... of the weight-gradient kernel (conv_wgrad_wino.hip for 3x3 / padding 1 with both channel counts multiples of 32)."""
    return "conv_wgrad_wino_f32" if (ks == 3 and pad == 1 and lib.aesr_conv2d_wgrad_up2_supported(int(cin), int(cout))) else "conv_wgrad_f32"


def _pb(kind, flops, nbytes=0.0):
    """``kind``: a kernel label, or ("wino" | "wgrad", layer arguments...) resolved through the library only while profiling.  ``nbytes``:
    algorithmic bytes of a bandwidth-bound call (its tensors read / written once)."""
    if PROFILER is not None:
        if isinstance(kind, tuple):
            kind = wino_kind(*kind[1:]) if kind[0] == "wino" else wgrad_kind(*kind[1:])
        PROFILER.begin(kind, flops, nbytes)


def _pe():
    if PROFILER is not None:
        PROFILER.end()


def _act_of(m):
    if isinstance(m, nn.LeakyReLU):
        return _hip.ACT_LRELU, float(m.negative_slope)
    if isinstance(m, nn.ReLU):
        return _hip.ACT_RELU, 0.0
    if isinstance(m, nn.Sigmoid):
        return _hip.ACT_SIGMOID, 0.0
    return None


class ConvStep:
    kind = "conv"

    def __init__(self, mod, act, slope):
        ks, st, pd = mod.kernel_size, mod.stride, mod.padding
        self.s2d = (ks == (2, 2) and st == (2, 2) and pd == (0, 0) and mod.dilation == (1, 1) and mod.groups == 1
                    and mod.in_channels % 4 == 0)
        if not self.s2d and (ks[0] != ks[1] or st != (1, 1) or pd[0] != pd[1] or ks[0] not in (1, 3) or mod.dilation != (1, 1)
                             or mod.groups != 1):
            raise NotImplementedError("HIP conv path supports square 1x1/3x3 stride-1 and 2x2 stride-2 convolutions, got %r" % (mod,))
        self.mod, self.act, self.slope = mod, act, slope
        self.cin, self.cout, self.ks, self.pad = mod.in_channels, mod.out_channels, ks[0], pd[0]
        if self.s2d:
            # stride-2 2x2 conv == space-to-depth followed by a 1x1 conv over 4*Cin channels (networks/acai_vanilla_strided.py:19)
            if act != _hip.ACT_NONE:
                raise NotImplementedError("activation fused behind a stride-2 convolution")
            self.cin_full, self.cin, self.ks, self.pad = mod.in_channels, 4 * mod.in_channels, 1, 0
        self.w1 = None          # s2d: the equivalent [Cout, 4Cin, 1, 1] filter
        self.packed = None      # forward operand (implicit GEMM packing)
        self.packed_t = None    # data-gradient operand
        self.packed_w = None    # forward operand, Winograd-transformed (U = G g G^T)
        self.packed_wt = None   # data-gradient operand, Winograd-transformed
        self.flipped = None     # Cout == 1: the flipped filter [9][Cin] of the data gradient (aesr_conv2d_cout1_dgrad_pre)
        self.packed_epoch = -1
        self.wino_fwd = (not self.s2d) and wino_ok(self.cin, self.cout, self.ks, self.pad, 0)
        self.wino_dgrad = (not self.s2d) and wino_ok(self.cin, self.cout, self.ks, self.pad, 1)
        self.in_up2 = False     # the nearest Upsample(x2) in front of this convolution is folded into its kernels (compile_steps)

    @property
    def mfma_fwd(self):
        return self.cin % 4 == 0

    def out_hw(self, h, w):
        if self.s2d:
            return h // 2, w // 2
        return h + 2 * self.pad - self.ks + 1, w + 2 * self.pad - self.ks + 1

    def weight_for_kernels(self):
        """[Cout,Cin,K,K] filter the kernels see (s2d: channels ordered (ky, kx, c) like aesr_space_to_depth2)."""
        if not self.s2d:
            return self.mod.weight
        self.w1 = self.mod.weight.detach().permute(0, 2, 3, 1).reshape(self.cout, self.cin, 1, 1).contiguous()
        return self.w1


class StemConvStep:
    """Encoder stem ``Conv2d(1, Cs, 1, padding=p)`` folded into the 3x3 convolution that follows it with no non-linearity
    in between (networks/acai_vanilla.py:51,55): one 1 -> C1 3x3 "thin" convolution, the Cs-channel stem tensor is never
    written (aesr_stemconv_*)."""
    kind = "stemconv"

    def __init__(self, stem, conv):
        self.stem, self.mod = stem.mod, conv.mod
        self.act, self.slope = conv.act, conv.slope
        self.cs, self.cout, self.stem_pad = stem.cout, conv.cout, stem.pad
        self.cin, self.ks, self.pad = 1, 3, 1
        self.s2d = False
        self.folded = None
        self.folded_epoch = -1

    def out_hw(self, h, w):
        return h + 2 * self.stem_pad, w + 2 * self.stem_pad


def _thin_channels(c):
    """Channel counts the bandwidth-bound "thin" kernels take: 4 * 2^k <= 256."""
    return 4 <= c <= 256 and c % 4 == 0 and ((c // 4) & (c // 4 - 1)) == 0


def fuse_stem(steps):
    """steps with the leading (1x1 single-channel stem, 3x3 pad-1 conv) pair replaced by one StemConvStep, or None."""
    if len(steps) < 2 or steps[0].kind != "conv" or steps[1].kind != "conv":
        return None
    st, cv = steps[0], steps[1]
    ok = (st.cin == 1 and st.ks == 1 and not st.s2d and st.act == _hip.ACT_NONE and st.mod.bias is not None
          and cv.ks == 3 and cv.pad == 1 and not cv.s2d and cv.cin == st.cout and _thin_channels(cv.cout))
    if not ok:
        return None
    return [StemConvStep(st, cv)] + list(steps[2:])


class BnStep:
    kind = "bn"

    def __init__(self, mod, mode):
        self.mod, self.mode, self.c = mod, mode, mod.num_features
        self.fold_up = False    # mode BN_UP whose upsampling the next convolution does in its loaders: apply at the input size

    @property
    def run_mode(self):
        return _hip.BN_NONE if self.fold_up else self.mode

    def out_hw(self, h, w):
        if self.fold_up:
            return h, w
        if self.mode == _hip.BN_POOL:
            return h // 2, w // 2
        if self.mode == _hip.BN_UP:
            return 2 * h, 2 * w
        return h, w


class ResampleStep:
    """Stand-alone AvgPool2d(2) / Upsample(x2, nearest | bilinear): not behind a BatchNorm (use_batchnorm=False stacks), or
    the bilinear form of networks/ae_standard.py:68 which the BatchNorm kernels do not fuse."""
    kind = "resample"

    def __init__(self, mode):
        self.mode = mode

    def out_hw(self, h, w):
        return (h // 2, w // 2) if self.mode == _hip.RS_POOL else (2 * h, 2 * w)


def _resample_of(m):
    if isinstance(m, nn.AvgPool2d):
        ks = m.kernel_size if isinstance(m.kernel_size, tuple) else (m.kernel_size, m.kernel_size)
        st = m.stride if isinstance(m.stride, tuple) else (m.stride, m.stride)
        if ks != (2, 2) or st != (2, 2) or m.padding not in (0, (0, 0)):
            raise NotImplementedError("only AvgPool2d(2) is lowered (got %r)" % (m,))
        return ResampleStep(_hip.RS_POOL)
    if isinstance(m, nn.Upsample):
        sf = m.scale_factor
        if sf not in (2, 2.0, (2, 2), (2.0, 2.0)):
            raise NotImplementedError("only Upsample(scale_factor=2) is lowered (got %r)" % (m,))
        if m.mode == "nearest":
            return ResampleStep(_hip.RS_NEAREST)
        if m.mode == "bilinear" and not m.align_corners:
            return ResampleStep(_hip.RS_BILINEAR)
        raise NotImplementedError("Upsample mode %s (align_corners=%s) has no HIP lowering" % (m.mode, m.align_corners))
    return None


def compile_steps(seq):
    mods = list(seq)
    steps, i = [], 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.Conv2d):
            act, slope = _hip.ACT_NONE, 0.0
            if i + 1 < len(mods) and _act_of(mods[i + 1]) is not None:
                act, slope = _act_of(mods[i + 1])
                i += 1
            steps.append(ConvStep(m, act, slope))
        elif isinstance(m, nn.BatchNorm2d):
            mode = _hip.BN_NONE
            if i + 1 < len(mods) and isinstance(mods[i + 1], nn.AvgPool2d):
                mode = _hip.BN_POOL
                i += 1
            elif i + 1 < len(mods) and isinstance(mods[i + 1], nn.Upsample) and mods[i + 1].mode == "nearest":
                _resample_of(mods[i + 1])         # validates the scale factor
                mode = _hip.BN_UP
                i += 1
            steps.append(BnStep(m, mode))
        elif _resample_of(m) is not None:
            steps.append(_resample_of(m))
        else:
            raise NotImplementedError("no HIP lowering for %r at position %d (use_batchnorm=False stacks are not "
                                      "covered)" % (m, i))
        i += 1
    # nearest Upsample(x2) between a BatchNorm and a 3x3 convolution is folded into that convolution's Winograd kernels (forward
    # and weight-gradient loaders read (y/2, x/2) of the half-resolution tensor, the data gradient stores 2x2 block sums): the
    # upsampled tensor and its gradient never exist (aesr_conv2d_wino_fwd_up2 / _dgrad_sum2, aesr_conv2d_wgrad_up2)
    for k, s in enumerate(steps[:-1]):
        nxt = steps[k + 1]
        if (FOLD_UPSAMPLE and s.kind == "bn" and s.mode == _hip.BN_UP and nxt.kind == "conv" and nxt.wino_fwd and nxt.wino_dgrad
                and lib.aesr_conv2d_wgrad_up2_supported(nxt.cin, nxt.cout)):
            s.fold_up, nxt.in_up2 = True, True
    for k, s in enumerate(steps):
        if s.kind == "bn" and (k == 0 or steps[k - 1].kind != "conv"):
            raise NotImplementedError("BatchNorm must follow a convolution")
        if s.kind == "resample" and (k == 0 or steps[k - 1].kind not in ("conv", "bn", "stemconv")):
            raise NotImplementedError("a pooling / upsampling step must follow a convolution or a BatchNorm")
    return steps


def _empty(shape, like, dtype=torch.float32):
    return torch.empty(shape, device=like.device, dtype=dtype)


class SequentialRunner:
    """Executes one compiled stack.  ``weights_epoch`` must be bumped whenever parameters change in place."""

    def __init__(self, seq):
        self.seq = seq
        self.steps = compile_steps(seq)
        self.steps_fused = fuse_stem(self.steps) if FUSE_STEM else None
        self.weights_epoch = 0
        self.params = [p for p in seq.parameters()]

    def mark_weights_dirty(self):
        """Host counters only (no launch).  Every state a kernel may have written through raw pointers counts as changed: the parameters
        (packed operands, folded stem) and the BatchNorm running statistics (eval-mode scale / shift cache) -- a replayed step graph
        updates both without any tensor version or Python-side counter moving, so the trainer calls this after every replay."""
        self.weights_epoch += 1
        self._bn_updates = getattr(self, "_bn_updates", 0) + 1

    # ---- weight preparation ---------------------------------------------------------------------------------
    def _prep_jobs(self, steps):
        """(jobs, commits) for every parameter-side operand of ``steps`` that is stale (after an optimizer step: all of them): the
        packed MFMA operands of the convolutions -- each direction in the form of the kernel that will run it: Winograd where it
        applies, implicit GEMM otherwise --, the flipped filter of a single-output-channel convolution's data gradient and the
        folded encoder stem.  ``commits``: closures that mark the operands fresh once the launch has been enqueued."""
        jobs, commits = [], []
        for s in steps:
            if s.kind == "stemconv":
                ws, w1 = s.stem.weight, s.mod.weight
                epoch = (self.weights_epoch, ws._version, s.stem.bias._version, w1._version, ws.data_ptr(), w1.data_ptr())
                if s.folded_epoch == epoch:
                    continue
                _hip.require_gpu_tensor(w1, "conv weight")
                n = lib.aesr_stemconv_folded_floats(s.cout)
                if s.folded is None or s.folded.numel() != n:
                    s.folded = _empty((n,), w1)
                jobs.append(_hip.PrepJob(w1.data_ptr(), ws.data_ptr(), s.stem.bias.data_ptr(), s.folded.data_ptr(), _hip.PREP_STEM_FOLD,
                                         s.cout, s.cs, 3, 0))
                commits.append(lambda s=s, epoch=epoch: setattr(s, "folded_epoch", epoch))
                continue
            if s.kind != "conv":
                continue
            w = s.mod.weight
            epoch = (self.weights_epoch, w._version, w.data_ptr())
            if s.packed_epoch == epoch:
                continue
            _hip.require_gpu_tensor(w, "conv weight")
            wk = s.weight_for_kernels()
            forms = ((0, "packed", s.cin % 4 == 0 and not s.wino_fwd, _hip.PREP_PACK), (1, "packed_t", s.cout % 4 == 0 and not s.wino_dgrad, _hip.PREP_PACK),
                     (0, "packed_w", s.wino_fwd, _hip.PREP_WINO_PACK), (1, "packed_wt", s.wino_dgrad, _hip.PREP_WINO_PACK))
            for transpose, attr, ok, kind in forms:
                if not ok:
                    continue
                n = (lib.aesr_conv2d_wino_packed_floats(s.cout, s.cin, transpose) if kind == _hip.PREP_WINO_PACK
                     else lib.aesr_conv2d_packed_floats(s.cout, s.cin, s.ks, transpose))
                buf = getattr(s, attr)
                if buf is None or buf.numel() != n:
                    buf = _empty((n,), wk)
                    setattr(s, attr, buf)
                jobs.append(_hip.PrepJob(wk.data_ptr(), None, None, buf.data_ptr(), kind, s.cout, s.cin, s.ks, transpose))
            if s.cout == 1 and s.ks == 3 and s.pad == 1 and _thin_channels(s.cin) and not s.s2d:
                if s.flipped is None or s.flipped.numel() != 9 * s.cin:
                    s.flipped = _empty((9 * s.cin,), wk)
                jobs.append(_hip.PrepJob(wk.data_ptr(), None, None, s.flipped.data_ptr(), _hip.PREP_COUT1_FLIP, 1, s.cin, 3, 0))
            commits.append(lambda s=s, epoch=epoch: setattr(s, "packed_epoch", epoch))
        return jobs, commits

    def _ensure_prepared(self, steps):
        prepare_weights([(self, steps)])

    def train_steps(self):
        """The step list a training pass runs (the stem-folded one when it exists): what ``prepare_weights`` readies ahead of the pass."""
        return self.steps_fused if self.steps_fused is not None else self.steps

    def forward(self, x, nstart, train, save, fused=True, first=0, last=None, raw_last=False):
        """x: NHWC fp32 [N,H,W,C]; nstart: group boundaries (len G+1).  Returns (out, saved, steps): ``steps`` is the
        compiled list that ran (the stem-folded one unless ``fused`` is False, e.g. when the input needs a gradient).
        ``first`` / ``last``: run only steps [first, last) of the UNFUSED list (inference: the decoder's first convolution apart from
        the rest); ``raw_last``: the last step that runs is a convolution and leaves out its activation (pre-activations)."""
        _hip.require_gpu_tensor(x, "input")
        N, H, W, C = x.shape
        G = len(nstart) - 1
        saved = []
        cur = x
        partial = first != 0 or last is not None
        steps = self.steps_fused if (fused and self.steps_fused is not None and not partial) else self.steps
        self._ensure_prepared(steps)
        if partial:
            if save:
                raise RuntimeError("a partial pass keeps nothing for a backward pass")
            steps = steps[first:last]
            if raw_last and (not steps or steps[-1].kind != "conv"):
                raise RuntimeError("raw_last needs a convolution as the last step of the range")
        skip = False
        for idx, s in enumerate(steps):
            if skip:                # a BatchNorm that went into the epilogue of the convolution in front of it (eval mode)
                skip = False
                continue
            if s.kind == "stemconv":
                if C != 1:
                    raise RuntimeError("channel mismatch: tensor has %d channels, the stem expects 1" % C)
                Ho, Wo = s.out_hw(H, W)
                out = _empty((N, Ho, Wo, s.cout), x)
                _pb("thin_expand", 0.0, 4.0 * (cur.numel() + out.numel()))
                check(lib.aesr_stemconv_fwd(ptr(cur), ptr(s.folded), ptr(s.mod.bias), ptr(out), N, H, W, s.cout, s.stem_pad,
                                            s.act, s.slope, stream()), "aesr_stemconv_fwd")
                _pe()
                if save:
                    saved.append((cur, out))
                if TRACE is not None:
                    TRACE.append((self.seq, s.mod, s.act, out))
                cur, H, W, C = out, Ho, Wo, s.cout
            elif s.kind == "conv":
                act_k = _hip.ACT_NONE if (raw_last and s is steps[-1]) else s.act
                if s.s2d:
                    if C != s.cin_full:
                        raise RuntimeError("channel mismatch: tensor has %d channels, conv expects %d" % (C, s.cin_full))
                    xs = _empty((N, H // 2, W // 2, 4 * C), x)
                    check(lib.aesr_space_to_depth2(ptr(cur), ptr(xs), N, H, W, C, stream()), "aesr_space_to_depth2")
                    full_hw = (H, W)
                    cur, H, W, C = xs, H // 2, W // 2, 4 * C
                if C != s.cin:
                    raise RuntimeError("channel mismatch: tensor has %d channels, conv expects %d" % (C, s.cin))
                if s.in_up2:
                    H, W = 2 * H, 2 * W                    # the convolution's size; ``cur`` stays at half resolution
                Ho, Wo = (H, W) if s.s2d else s.out_hw(H, W)
                out = _empty((N, Ho, Wo, s.cout), x)
                bias = s.mod.bias
                if s.in_up2:
                    _pb(("wino", N, Ho, Wo, s.cin, s.cout, 0), 2.0 * N * Ho * Wo * s.cout * 9 * s.cin)
                    check(lib.aesr_conv2d_wino_fwd_up2(ptr(cur), ptr(s.packed_w), ptr(bias), ptr(out), N, H, W, s.cin, s.cout, act_k,
                                                       s.slope, stream()), "aesr_conv2d_wino_fwd_up2")
                    _pe()
                elif s.cout == 1 and s.ks == 3 and s.pad == 1 and s.cin % 4 == 0:
                    _pb("thin_collapse", 0.0, 4.0 * (cur.numel() + out.numel()))
                    check(lib.aesr_conv2d_cout1_fwd(ptr(cur), ptr(s.mod.weight), ptr(bias), ptr(out), N, H, W, s.cin, act_k,
                                                    s.slope, stream()), "aesr_conv2d_cout1_fwd")
                    _pe()
                elif (s.wino_fwd and FUSE_EVAL_BN and not train and not save and G == 1 and idx + 1 < len(steps) and steps[idx + 1].kind == "bn"
                      and steps[idx + 1].run_mode in (_hip.BN_NONE, _hip.BN_POOL) and steps[idx + 1].mod.running_mean is not None
                      and not (raw_last and s is steps[-1]) and (steps[idx + 1].run_mode == _hip.BN_NONE or (H >= 2 and W >= 2))
                      and lib.aesr_conv2d_wino_fwd_bn_supported(N, H, W, s.cin, s.cout)):
                    # eval mode: BatchNorm is a per-channel affine of the running statistics (cached per layer state) -- it and the pooling
                    # behind it go into this convolution's epilogue; the activation tensor in between is never written
                    nxt = steps[idx + 1]
                    st = self._bn_forward_stats(nxt.mod, cur, N, Ho, Wo, s.cout, nstart, False)
                    Hb, Wb = nxt.out_hw(Ho, Wo)
                    out = _empty((N, Hb, Wb, s.cout), x)
                    _pb(("wino", N, Ho, Wo, s.cin, s.cout, 0), 2.0 * N * Ho * Wo * s.cout * 9 * s.cin)
                    check(lib.aesr_conv2d_wino_fwd_bn(ptr(cur), ptr(s.packed_w), ptr(bias), ptr(st["scale"]), ptr(st["shift"]), ptr(out), N, H, W,
                                                      s.cin, s.cout, act_k, s.slope, int(nxt.run_mode == _hip.BN_POOL), stream()),
                          "aesr_conv2d_wino_fwd_bn")
                    _pe()
                    Ho, Wo, skip = Hb, Wb, True
                elif s.wino_fwd:
                    _pb(("wino", N, Ho, Wo, s.cin, s.cout, 0), 2.0 * N * Ho * Wo * s.cout * 9 * s.cin)
                    ws, nws = wino_workspace(cur, N, H, W, s.cin, s.cout, 0)
                    check(lib.aesr_conv2d_wino_fwd_ws(ptr(cur), ptr(s.packed_w), ptr(bias), ptr(out), ptr(ws), nws, N, H, W, s.cin, s.cout,
                                                      act_k, s.slope, stream()), "aesr_conv2d_wino_fwd_ws")
                    _pe()
                elif s.mfma_fwd:
                    _pb("conv_igemm_f32", 2.0 * N * Ho * Wo * s.cout * s.ks * s.ks * s.cin)
                    nws = lib.aesr_conv2d_workspace_floats(N, H, W, s.cin, s.cout, s.ks, s.pad)    # > 0: few, deep work items
                    ws = _empty((nws,), x) if nws else None
                    check(lib.aesr_conv2d_fwd_ws(ptr(cur), ptr(s.packed), ptr(bias), ptr(out), ptr(ws), N, H, W, s.cin, s.cout,
                                                 s.ks, s.pad, act_k, s.slope, stream()), "aesr_conv2d_fwd_ws")
                    _pe()
                elif s.cin <= 4:
                    check(lib.aesr_conv2d_smallcin_fwd(ptr(cur), ptr(s.mod.weight), ptr(bias), None, ptr(out), N, H, W,
                                                       s.cin, s.cout, s.ks, s.pad, act_k, _hip.ACT_NONE, s.slope, 0, 0,
                                                       None, None, stream()), "aesr_conv2d_smallcin_fwd")
                else:
                    raise NotImplementedError("conv with Cin=%d (neither <=4 nor a multiple of 4)" % s.cin)
                if save:
                    saved.append((cur, out, full_hw) if s.s2d else (cur, out))
                if TRACE is not None and not skip:
                    TRACE.append((self.seq, s.mod, act_k, out))
                cur, H, W, C = out, Ho, Wo, s.cout
            elif s.kind == "resample":
                if C % 4 != 0:
                    raise NotImplementedError("stand-alone pooling / upsampling of %d channels (needs a multiple of 4)" % C)
                Ho, Wo = s.out_hw(H, W)
                out = _empty((N, Ho, Wo, C), x)
                check(lib.aesr_resample2_fwd(ptr(cur), ptr(out), N, H, W, C, s.mode, stream()), "aesr_resample2_fwd")
                if save:
                    saved.append((cur,))
                cur, H, W = out, Ho, Wo
            else:
                bn = s.mod
                Ho, Wo = s.out_hw(H, W)
                out = _empty((N, Ho, Wo, C), x)
                _pb("bn_fwd", 0.0, 4.0 * (cur.numel() + out.numel()))
                st = self._bn_forward_stats(bn, cur, N, H, W, C, nstart, train, out, s.run_mode)
                if not st.pop("applied", False):
                    check(lib.aesr_bn_apply(ptr(cur), ptr(st["scale"]), ptr(st["shift"]), ptr(out), N, H, W, C, s.run_mode, G,
                                            _hip.int_array(nstart), stream()), "aesr_bn_apply")
                _pe()
                if save:
                    saved.append((cur, st))
                cur, H, W = out, Ho, Wo
        return cur, saved, steps

    def max_elems_per_image(self, H, W, C, first=0, last=None):
        """Largest activation tensor (elements per image) a pass over ``steps[first:last]`` touches for an H x W x C input: what decides
        how many images fit one pass under the kernels' 32-bit offsets (aesr_launch_conv_wino refuses tensors of 469 M elements)."""
        return self.trace_shapes(H, W, C, first, last)[0]

    def trace_shapes(self, H, W, C, first=0, last=None):
        """(largest activation tensor in elements per image, (H, W, C) of the output) of a pass over ``steps[first:last]``."""
        best = H * W * C
        for s in self.steps[first:last]:
            if s.kind == "conv":
                if s.s2d:
                    H, W = H // 2, W // 2
                elif s.in_up2:
                    H, W = 2 * H, 2 * W
                    H, W = s.out_hw(H, W)
                else:
                    H, W = s.out_hw(H, W)
                C = s.cout
            elif s.kind == "stemconv":
                H, W = s.out_hw(H, W)
                C = s.cout
            else:
                H, W = s.out_hw(H, W)
            best = max(best, H * W * C)
        return best, (H, W, C)

    def _bn_barrier(self, dev):
        """Grid-barrier state of this runner's one-launch BatchNorm kernels (csrc/bn_fused.hip): zeroed once, then owned by the kernels.
        Created by an eager step (a tensor born inside a graph capture would live in the graph's pool and be zeroed by every replay)."""
        t = self.__dict__.get("_bn_bar")
        if t is None or t.device != torch.device(dev):
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("one-launch BatchNorm: run the pass eagerly once before capturing it into a HIP graph")
            t = self._bn_bar = torch.zeros((int(lib.aesr_bn_fused1_barrier_words()),), device=dev, dtype=torch.int32)
        return t

    sync_p2p = None     # optional parallel.PeerExchange: the SyncBN exchange inside the one-launch BatchNorm kernels (AESR_SYNCBN=p2p)

    def _p2p_fits(self, N, H, W, C, run_mode, G, backward):
        """Data parallel with the peer exchange: does this BatchNorm call take the one-launch kernel?  Decided on the LARGEST shard of
        the step (uneven shards), so that every rank answers the same -- a rank in the kernel would wait for a rank in the all-reduce."""
        p = self.sync_p2p
        if p is None or not bn_fused_enabled():
            return False
        nmax = max(N, int(round(N * p.nscale)))
        # big layers are not for the one-launch kernel: it streams at 1.7-1.8 TB/s, and with the 18.9 MB second BatchNorm of a 2-triplet C4 shard
        # (6 x 111 x 111 x 64) in it the rank step read 3.13 instead of 2.65 ms unprofiled -- not as kernel time under rocprofv3, where kernels do
        # not overlap at their boundaries: its 256 workgroups of ~140 KB LDS each can only start once the previous kernel has left EVERY CU, and the
        # early ones spin at the grid barrier meanwhile (profiles/r05_p2p_c4_layer_cap.txt).  Above AESR_P2P_MAX_MB (12) a call keeps the all-reduce
        # form.  Decided on the largest shard of the step, like the fit, so that every rank answers alike.
        if float(nmax) * H * W * C * 4.0 > float(os.environ.get("AESR_P2P_MAX_MB", "12")) * 1e6:
            return False
        return bool(lib.aesr_bn_fused1_supported(nmax, H, W, C, run_mode, G, backward))

    sync_bn = None      # optional callable(sums[G,2,C] double) -> all-reduced in place across ranks (data parallel SyncBN)
    count_scale = 1.0   # data parallel: global / local sub-batch size (B_global / B_local of this rank)

    def _bn_forward_stats(self, bn, y, N, H, W, C, nstart, train, out=None, run_mode=0):
        """Statistics of a BatchNorm call -> {mean, invstd, scale, shift, counts}.  Data parallel (``sync_bn``) with ``out`` given: the
        finalize runs inside the apply launch (``"applied": True`` in the result -- the caller must not apply again)."""
        G = len(nstart) - 1
        dev = y.device
        st = {k: torch.empty((G, C), device=dev, dtype=torch.float32) for k in ("mean", "invstd", "scale", "shift")}
        use_batch = train or bn.running_mean is None
        momentum = 0.1 if bn.momentum is None else float(bn.momentum)
        update = bool(train and bn.track_running_stats and bn.running_mean is not None)
        if update:
            self._bn_updates = getattr(self, "_bn_updates", 0) + 1       # the kernels write the running statistics through raw pointers
        if not use_batch:
            # eval mode: scale / shift follow from the running statistics alone -- computed once per state of the layer, not once per call
            # (slice synthesis runs 6 BatchNorm calls per volume: 6 launches of a few hundred threads each)
            ver = lambda t: (-1, 0) if t is None else (t._version, t.data_ptr())         # affine=False: no weight / bias
            key = (self.weights_epoch, getattr(self, "_bn_updates", 0), ver(bn.weight), ver(bn.bias), ver(bn.running_mean), ver(bn.running_var),
                   G, str(dev))
            cache = self.__dict__.setdefault("_eval_bn", {})
            hit = cache.get(id(bn))
            if hit is not None and hit[0] == key and not torch.cuda.is_current_stream_capturing():
                return dict(hit[1])
        sums = counts = None
        if use_batch:
            partial = torch.empty((G * _hip.BN_NWG * 2 * C,), device=dev, dtype=torch.float32)
            counts = [float((nstart[g + 1] - nstart[g]) * H * W) * self.count_scale for g in range(G)]   # host values
            st["counts"] = counts
            if (self.sync_bn is None and out is not None and bn_fused_enabled() and bn_fused_pays(N, H, W, C)
                    and lib.aesr_bn_fused1_supported(N, H, W, C, run_mode, G, 0)):
                # small batch: statistics, finalize and apply in ONE launch, the layer resident in LDS in between
                ws = torch.empty((lib.aesr_bn_fused1_workspace_floats(C, G),), device=dev, dtype=torch.float32)
                check(lib.aesr_bn_fused1_fwd(ptr(y), ptr(out), ptr(ws), ptr(self._bn_barrier(dev)), _hip.double_array(counts), ptr(bn.weight),
                                             ptr(bn.bias), ptr(bn.running_mean), ptr(bn.running_var), ptr(bn.num_batches_tracked),
                                             ptr(st["mean"]), ptr(st["invstd"]), ptr(st["scale"]), ptr(st["shift"]), N, H, W, C, run_mode, G,
                                             _hip.int_array(nstart), momentum, float(bn.eps), int(update), stream()), "aesr_bn_fused1_fwd")
                st["applied"] = True
                return st
            if self.sync_bn is not None and out is not None and self._p2p_fits(N, H, W, C, run_mode, G, 0):
                # data parallel, peer exchange: the same ONE launch; workgroup 0 writes this rank's sums into every rank's region and
                # every workgroup adds all ranks' sums in rank order before it normalises
                p = self.sync_p2p
                ws = torch.empty((lib.aesr_bn_fused1_workspace_floats(C, G),), device=dev, dtype=torch.float32)
                check(lib.aesr_bn_fused1_fwd_p2p(ptr(y), ptr(out), ptr(ws), ptr(self._bn_barrier(dev)), _hip.double_array(counts), ptr(bn.weight),
                                                 ptr(bn.bias), ptr(bn.running_mean), ptr(bn.running_var), ptr(bn.num_batches_tracked),
                                                 ptr(st["mean"]), ptr(st["invstd"]), ptr(st["scale"]), ptr(st["shift"]), N, H, W, C, run_mode, G,
                                                 _hip.int_array(nstart), momentum, float(bn.eps), int(update), p.peers, p.world, p.rank,
                                                 p.next_slot(), ptr(p.gen), stream()), "aesr_bn_fused1_fwd_p2p")
                st["applied"] = True
                return st
            if self.sync_bn is None:        # single process: statistics -> finalize without the sums round trip
                check(lib.aesr_bn_stats_finalize(ptr(y), ptr(partial), _hip.double_array(counts), ptr(bn.weight), ptr(bn.bias),
                                                 ptr(bn.running_mean), ptr(bn.running_var), ptr(bn.num_batches_tracked),
                                                 ptr(st["mean"]), ptr(st["invstd"]), ptr(st["scale"]), ptr(st["shift"]), H * W, C,
                                                 G, _hip.int_array(nstart), momentum, float(bn.eps), int(update), stream()),
                      "aesr_bn_stats_finalize")
                return st
            sums = torch.empty((G, 2, C), device=dev, dtype=torch.float64)
            check(lib.aesr_bn_stats(ptr(y), ptr(partial), ptr(sums), H * W, C, G, _hip.int_array(nstart), stream()),
                  "aesr_bn_stats")
            self.sync_bn(sums)
            if out is not None and lib.aesr_bn_fused_supported(C, G):
                st["counts"] = counts
                check(lib.aesr_bn_finalize_apply(ptr(sums), _hip.double_array(counts), ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean),
                                                 ptr(bn.running_var), ptr(bn.num_batches_tracked), ptr(st["mean"]), ptr(st["invstd"]),
                                                 ptr(st["scale"]), ptr(st["shift"]), ptr(y), ptr(out), N, H, W, C, run_mode, G,
                                                 _hip.int_array(nstart), momentum, float(bn.eps), int(update), stream()),
                      "aesr_bn_finalize_apply")
                st["applied"] = True
                return st
        st["counts"] = counts
        check(lib.aesr_bn_finalize(ptr(sums), _hip.double_array(counts) if counts else None, ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean),
                                   ptr(bn.running_var), ptr(bn.num_batches_tracked), ptr(st["mean"]), ptr(st["invstd"]),
                                   ptr(st["scale"]), ptr(st["shift"]), C, G, momentum, float(bn.eps), int(use_batch),
                                   int(update), stream()), "aesr_bn_finalize")
        if not use_batch and not torch.cuda.is_current_stream_capturing():
            self.__dict__.setdefault("_eval_bn", {})[id(bn)] = (key, dict(st))
        return st

    # ---- backward --------------------------------------------------------------------------------------------
    @staticmethod
    def _grad_dst(p, grads):
        """Where a parameter gradient is written: straight into ``p.grad`` when that buffer is known to be freshly zeroed
        (HipAdam.zero_grad marks it) -- autograd then gets ``None`` and no accumulation kernel runs -- else a temporary."""
        if p.grad is not None and getattr(p, "_aesr_grad_fresh", False) and p.grad.is_contiguous():
            p._aesr_grad_fresh = False
            grads[p] = None
            return p.grad
        t = torch.empty_like(p)
        grads[p] = t
        return t

    def backward(self, gout, saved, nstart, ngrad, need_input_grad, steps=None):
        """gout: NHWC gradient of the pass output (all N images; only the first ``ngrad`` are used).
        Returns (dx or None, {param: grad})."""
        steps = self.steps if steps is None else steps
        grads = {}
        G = 0
        while G < len(nstart) - 1 and nstart[G + 1] <= ngrad:
            G += 1
        if G == 0 or nstart[G] != ngrad:
            raise RuntimeError("the images that need gradients must be whole leading statistic groups")
        ns = list(nstart[:G + 1])
        g = gout[:ngrad]
        if not g.is_contiguous():
            g = g.contiguous()
        reduce_jobs = []          # (job, workspace, dw, db, direct): the tensors stay referenced until the reduction has been enqueued
        g = self._backward_steps(steps, saved, g, ngrad, need_input_grad, grads, G, ns, reduce_jobs)
        if _DEFERRED is not None:
            # only reductions that write straight into ``p.grad`` may wait for the end of the sweep (ONE launch for all passes,
            # deferred_wgrad_reductions).  A destination that is a temporary goes back to autograd when this function returns, and
            # autograd adds it to ``p.grad`` at once: it must be complete NOW (a parameter differentiated by a second pass of the
            # sweep, an optimizer other than HipAdam, gradients that were not freshly zeroed)
            _DEFERRED.extend(j for j in reduce_jobs if j[4])
            flush_wgrad_reductions([j for j in reduce_jobs if not j[4]])
        else:
            flush_wgrad_reductions(reduce_jobs)
        return g, grads

    def _backward_steps(self, steps, saved, g, ngrad, need_input_grad, grads, G, ns, reduce_jobs):
        for k in range(len(steps) - 1, -1, -1):
            s = steps[k]
            if s.kind == "stemconv":
                if need_input_grad:
                    raise RuntimeError("the stem-folded pass has no input gradient (run the pass with fused=False)")
                xin = saved[k][0]
                _, H, W, _ = xin.shape
                dws, dbs = self._grad_dst(s.stem.weight, grads), self._grad_dst(s.stem.bias, grads)
                dw1 = self._grad_dst(s.mod.weight, grads)
                db1 = self._grad_dst(s.mod.bias, grads) if s.mod.bias is not None else None
                ws = _empty((lib.aesr_stemconv_workspace_floats(s.cout),), g)
                _pb("thin_reduce", 0.0, 4.0 * (ngrad * H * W + g.numel()))
                check(lib.aesr_stemconv_wgrad(ptr(xin), ptr(g), ptr(s.stem.weight), ptr(s.stem.bias), ptr(s.mod.weight),
                                              ptr(dws), ptr(dbs), ptr(dw1), ptr(db1), ptr(ws), ngrad, H, W, s.cs, s.cout,
                                              s.stem_pad, stream()), "aesr_stemconv_wgrad")
                _pe()
                g = None
                break
            if s.kind == "conv":
                xin, yout = saved[k][0], saved[k][1]
                N, H, W, _ = xin.shape
                N = ngrad
                if s.in_up2:
                    H, W = 2 * H, 2 * W                    # the convolution's size; xin is the half-resolution tensor
                Ho, Wo = (H, W) if s.s2d else s.out_hw(H, W)
                if k == len(steps) - 1 and s.act != _hip.ACT_NONE:
                    dpre = torch.empty_like(g)
                    check(lib.aesr_act_bwd(ptr(g), ptr(yout), ptr(dpre), g.numel(), s.act, s.slope, stream()), "aesr_act_bwd")
                    g = dpre
                # -- weight / bias gradient
                dw_final = self._grad_dst(s.mod.weight, grads)
                dw = torch.empty((s.cout, s.cin, 1, 1), device=g.device) if s.s2d else dw_final
                db = self._grad_dst(s.mod.bias, grads) if s.mod.bias is not None else None
                if s.cin % 4 == 0 and s.cout % 4 == 0:
                    nws = lib.aesr_conv2d_wgrad_workspace_floats(N, H, W, s.cin, s.cout, s.ks, s.pad)
                    ws = _empty((nws,), g)
                    if s.s2d:        # its result is re-laid out right below: reduce at once
                        _pb(("wgrad", s.cin, s.cout, s.ks, s.pad), 2.0 * N * Ho * Wo * s.cout * s.ks * s.ks * s.cin)
                        check(lib.aesr_conv2d_wgrad(ptr(xin), ptr(g), ptr(dw), ptr(db), ptr(ws), N, H, W, s.cin, s.cout, s.ks,
                                                    s.pad, stream()), "aesr_conv2d_wgrad")
                        _pe()
                    else:
                        # partial slabs now; the slabs of all layers of this pass are summed by ONE launch at the end of the pass
                        _pb(("wgrad", s.cin, s.cout, s.ks, s.pad), 2.0 * N * Ho * Wo * s.cout * s.ks * s.ks * s.cin)
                        check(lib.aesr_conv2d_wgrad_partial(ptr(xin), ptr(g), ptr(ws), N, H, W, s.cin, s.cout, s.ks, s.pad,
                                                            int(s.in_up2), stream()), "aesr_conv2d_wgrad_partial")
                        _pe()
                        direct = grads[s.mod.weight] is None and (db is None or grads[s.mod.bias] is None)
                        reduce_jobs.append((_hip.WgradReduceJob(ws.data_ptr(), dw.data_ptr(), db.data_ptr() if db is not None else None,
                                                                N, H, W, s.cin, s.cout, s.ks, s.pad), ws, dw, db, direct))
                elif s.cin <= 4 and s.ks == 1 and db is not None:
                    ws = _empty((lib.aesr_small_wgrad_workspace_floats(s.cout * (s.cin + 1)),), g)
                    check(lib.aesr_conv2d_smallcin_wgrad(ptr(xin), ptr(g), ptr(dw), ptr(db), ptr(ws), N, H, W, s.cin, s.cout,
                                                         s.pad, stream()), "aesr_conv2d_smallcin_wgrad")
                elif s.cout == 1 and s.ks == 3 and s.pad == 1 and db is not None:
                    ws = _empty((lib.aesr_conv2d_cout1_workspace_floats(s.cin),), g)
                    _pb("thin_reduce", 0.0, 4.0 * (N * H * W * s.cin + g.numel()))
                    check(lib.aesr_conv2d_cout1_wgrad(ptr(xin), ptr(g), ptr(dw), ptr(db), ptr(ws), N, H, W, s.cin, stream()),
                          "aesr_conv2d_cout1_wgrad")
                    _pe()
                else:
                    raise NotImplementedError("no wgrad kernel for conv %d->%d k%d" % (s.cin, s.cout, s.ks))
                if s.s2d:       # [Cout, (ky,kx,c)] -> [Cout, c, ky, kx]
                    dw_final.copy_(dw.reshape(s.cout, 2, 2, s.cin_full).permute(0, 3, 1, 2))
                # -- data gradient (fused with the derivative of the activation that produced our input)
                if k == 0 and not need_input_grad:
                    g = None
                    break
                mask, mask_act, mslope = None, _hip.ACT_NONE, 0.0
                if k > 0 and steps[k - 1].kind in ("conv", "stemconv"):
                    mask, mask_act, mslope = saved[k - 1][1], steps[k - 1].act, steps[k - 1].slope
                    if mask_act == _hip.ACT_NONE:
                        mask = None
                dx = _empty((N, H // 2, W // 2, s.cin) if s.in_up2 else (N, H, W, s.cin), g)
                if s.in_up2:
                    # adjoint of the folded upsampling: 2x2 block sums of the data gradient, at half resolution (no mask: the
                    # producer is a BatchNorm)
                    _pb(("wino", N, H, W, s.cin, s.cout, 1), 2.0 * N * H * W * s.cin * 9 * s.cout)
                    check(lib.aesr_conv2d_wino_dgrad_sum2(ptr(g), ptr(s.packed_wt), ptr(dx), N, H, W, s.cin, s.cout, stream()),
                          "aesr_conv2d_wino_dgrad_sum2")
                    _pe()
                elif s.cin <= 4 and mask is None:
                    check(lib.aesr_conv2d_smallcin_dgrad(ptr(g), ptr(s.mod.weight), ptr(dx), N, H, W, s.cin, s.cout, s.ks,
                                                         s.pad, 0, None, stream()), "aesr_conv2d_smallcin_dgrad")
                elif s.wino_dgrad:
                    _pb(("wino", N, H, W, s.cin, s.cout, 1), 2.0 * N * H * W * s.cin * 9 * s.cout)
                    ws, nws = wino_workspace(g, N, H, W, s.cin, s.cout, 1)
                    check(lib.aesr_conv2d_wino_dgrad_ws(ptr(g), ptr(s.packed_wt), ptr(mask), ptr(dx), ptr(ws), nws, N, H, W, s.cin, s.cout,
                                                        mask_act, mslope, stream()), "aesr_conv2d_wino_dgrad_ws")
                    _pe()
                elif s.cout % 4 == 0:
                    _pb("conv_igemm_f32", 2.0 * N * H * W * s.cin * s.ks * s.ks * s.cout)
                    nws = lib.aesr_conv2d_dgrad_workspace_floats(N, H, W, s.cin, s.cout, s.ks, s.pad)
                    ws = _empty((nws,), g) if nws else None
                    check(lib.aesr_conv2d_dgrad_ws(ptr(g), ptr(s.packed_t), ptr(mask), ptr(dx), ptr(ws), N, H, W, s.cin, s.cout,
                                                   s.ks, s.pad, mask_act, mslope, stream()), "aesr_conv2d_dgrad_ws")
                    _pe()
                elif s.cout == 1 and s.ks == 3 and s.pad == 1 and _thin_channels(s.cin):
                    w = s.mod.weight
                    if s.flipped is not None and s.packed_epoch == (self.weights_epoch, w._version, w.data_ptr()):
                        # the flipped filter was made with the rest of the step's operands (prepare_weights): one launch
                        _pb("thin_expand", 0.0, 4.0 * (g.numel() + dx.numel() * (2 if mask is not None else 1)))
                        check(lib.aesr_conv2d_cout1_dgrad_pre(ptr(g), ptr(s.flipped), ptr(mask), ptr(dx), N, H, W, s.cin, mask_act, mslope,
                                                              stream()), "aesr_conv2d_cout1_dgrad_pre")
                        _pe()
                    else:
                        wsf = _empty((9 * s.cin,), g)
                        check(lib.aesr_conv2d_cout1_dgrad(ptr(g), ptr(w), ptr(mask), ptr(dx), ptr(wsf), N, H, W, s.cin,
                                                          mask_act, mslope, stream()), "aesr_conv2d_cout1_dgrad")
                elif s.cout <= 4:
                    # data gradient of a tiny-Cout conv == small-Cin forward conv with the flipped/transposed filter
                    check(lib.aesr_conv2d_smallcin_fwd(ptr(g), ptr(s.mod.weight), None, ptr(mask), ptr(dx), N, Ho, Wo, s.cout,
                                                       s.cin, s.ks, s.ks - 1 - s.pad, _hip.ACT_NONE, mask_act, mslope, 1, 0,
                                                       None, None, stream()), "aesr_conv2d_smallcin_fwd(dgrad)")
                else:
                    raise NotImplementedError("no dgrad kernel for conv %d->%d" % (s.cin, s.cout))
                if s.s2d:
                    if mask is not None:
                        raise NotImplementedError("activation mask in front of a stride-2 convolution")
                    fh, fw = saved[k][2]
                    dfull = _empty((N, fh, fw, s.cin_full), g)
                    check(lib.aesr_depth_to_space2(ptr(dx), ptr(dfull), N, fh, fw, s.cin_full, stream()), "aesr_depth_to_space2")
                    dx = dfull
                g = dx
            elif s.kind == "resample":
                xin = saved[k][0]
                _, H, W, C = xin.shape
                prev = steps[k - 1]
                mask_act, mslope = (prev.act, prev.slope) if prev.kind in ("conv", "stemconv") else (_hip.ACT_NONE, 0.0)
                dx = _empty((ngrad, H, W, C), g)
                check(lib.aesr_resample2_bwd(ptr(g), ptr(xin) if mask_act != _hip.ACT_NONE else None, ptr(dx), ngrad, H, W, C,
                                             s.mode, mask_act, mslope, stream()), "aesr_resample2_bwd")
                g = dx
            else:
                y, st = saved[k]
                prev = steps[k - 1]
                _, H, W, C = y.shape
                N = ngrad
                dev = y.device
                partial = torch.empty((G * _hip.BN_NWG * 2 * C,), device=dev, dtype=torch.float32)
                nsa = _hip.int_array(ns)
                coef = torch.empty((G, 2, C), device=dev, dtype=torch.float32)
                dgamma, dbeta = self._grad_dst(s.mod.weight, grads), self._grad_dst(s.mod.bias, grads)
                dpre = _empty((N, H, W, C), y)
                _pb("bn_bwd", 0.0, 4.0 * (g.numel() + 2.0 * N * H * W * C))
                if (self.sync_bn is None and bn_fused_enabled() and bn_fused_pays(N, H, W, C)
                        and lib.aesr_bn_fused1_supported(N, H, W, C, s.run_mode, G, 1)):
                    ws = torch.empty((lib.aesr_bn_fused1_workspace_floats(C, G),), device=dev, dtype=torch.float32)
                    check(lib.aesr_bn_fused1_bwd(ptr(g), ptr(y), ptr(st["mean"]), ptr(st["invstd"]), ptr(st["scale"]), ptr(ws),
                                                 ptr(self._bn_barrier(dev)), _hip.double_array(st["counts"][:G]), ptr(coef), ptr(dgamma), ptr(dbeta),
                                                 ptr(dpre), N, H, W, C, s.run_mode, prev.act, prev.slope, G, nsa, stream()), "aesr_bn_fused1_bwd")
                elif self.sync_bn is not None and self._p2p_fits(N, H, W, C, s.run_mode, G, 1):
                    p = self.sync_p2p
                    ws = torch.empty((lib.aesr_bn_fused1_workspace_floats(C, G),), device=dev, dtype=torch.float32)
                    check(lib.aesr_bn_fused1_bwd_p2p(ptr(g), ptr(y), ptr(st["mean"]), ptr(st["invstd"]), ptr(st["scale"]), ptr(ws),
                                                     ptr(self._bn_barrier(dev)), _hip.double_array(st["counts"][:G]), ptr(coef), ptr(dgamma),
                                                     ptr(dbeta), ptr(dpre), N, H, W, C, s.run_mode, prev.act, prev.slope, G, nsa, p.peers, p.world,
                                                     p.rank, p.next_slot(), ptr(p.gen), stream()), "aesr_bn_fused1_bwd_p2p")
                elif self.sync_bn is None:
                    check(lib.aesr_bn_bwd(ptr(g), ptr(y), ptr(st["mean"]), ptr(st["invstd"]), ptr(st["scale"]), ptr(partial),
                                          _hip.double_array(st["counts"][:G]), ptr(coef), ptr(dgamma), ptr(dbeta), ptr(dpre), N, H, W,
                                          C, s.run_mode, prev.act, prev.slope, G, nsa, stream()), "aesr_bn_bwd")
                else:
                    sums = torch.empty((G, 2, C), device=dev, dtype=torch.float64)
                    check(lib.aesr_bn_bwd_reduce(ptr(g), ptr(y), ptr(st["mean"]), ptr(st["invstd"]), ptr(partial), ptr(sums), N,
                                                 H, W, C, s.run_mode, G, nsa, stream()), "aesr_bn_bwd_reduce")
                    self.sync_bn(sums)
                    check(lib.aesr_bn_bwd_apply(ptr(g), ptr(y), ptr(st["mean"]), ptr(st["invstd"]), ptr(st["scale"]), ptr(sums),
                                                _hip.double_array(st["counts"][:G]), ptr(coef), ptr(dgamma), ptr(dbeta), ptr(dpre),
                                                N, H, W, C, s.run_mode, prev.act, prev.slope, G, nsa, stream()), "aesr_bn_bwd_apply")
                _pe()
                g = dpre
        return g


def prepare_weights(pairs):
    """ONE aesr_weight_prep_many launch (per 32 jobs) for everything stale in ``pairs`` = [(runner, step list), ...]: the operands of all
    networks of a training step at once (AEBaseTrainer calls this at the top of the step for encoder + decoder: one graph node instead
    of four to five); a pass calls it for its own list, which then finds nothing left to do."""
    jobs, commits = [], []
    for runner, steps in pairs:
        j, c = runner._prep_jobs(steps)
        jobs += j
        commits += c
    if jobs:
        arr = (_hip.PrepJob * len(jobs))(*jobs)
        _pb("prep_many", 0.0, 0.0)
        check(lib.aesr_weight_prep_many(arr, len(jobs), stream()), "aesr_weight_prep_many")
        _pe()
    for c in commits:
        c()


_DEFERRED = None      # list of pending weight-gradient reduction jobs while a deferred_wgrad_reductions() block is open

def flush_wgrad_reductions(jobs):
    """ONE aesr_conv2d_wgrad_reduce_many launch per 16 layers for the slab sets in ``jobs``."""
    for k in range(0, len(jobs), _hip.REDUCE_MAX_JOBS):
        part = jobs[k:k + _hip.REDUCE_MAX_JOBS]
        arr = (_hip.WgradReduceJob * len(part))(*[j[0] for j in part])
        _pb("wgrad_reduce_many", 0.0, 4.0 * sum(j[1].numel() + j[2].numel() for j in part))
        check(lib.aesr_conv2d_wgrad_reduce_many(arr, len(part), stream()), "aesr_conv2d_wgrad_reduce_many")
        _pe()


class deferred_wgrad_reductions(object):
    """``with deferred_wgrad_reductions(): loss.backward()``: the weight-gradient slab sets of EVERY pass of the backward sweep
    (decoder, then encoder) are summed by one launch when the block closes instead of one launch per pass.  Parameter gradients
    are complete only after the block."""

    def __enter__(self):
        global _DEFERRED
        self.outer = _DEFERRED
        if os.environ.get("AESR_DEFER_REDUCE", "1") != "0":
            _DEFERRED = []
        return self

    def __exit__(self, *exc):
        global _DEFERRED
        jobs, _DEFERRED = (_DEFERRED or []), self.outer
        if exc[0] is None:
            if self.outer is not None:
                self.outer.extend(jobs)
            else:
                flush_wgrad_reductions(jobs)
        return False


class _PassFn(torch.autograd.Function):
    """One pass of a compiled stack as a single autograd node: NHWC batch in, one logical-NCHW view of the NHWC output
    buffer per sub-batch out (so the caller never runs an autograd split / cat / layout copy on the activations)."""

    @staticmethod
    def forward(ctx, runner, nstart, ngrad, train, splits, x, *params):
        need = any(ctx.needs_input_grad) and ngrad > 0       # grad mode is off inside forward(); this is the truth
        ctx.set_materialize_grads(False)     # sub-batches without a gradient (the logging-only encoder pass) come back as None, not as zeros
        out, saved, steps = runner.forward(x.detach(), nstart, train, save=need, fused=not x.requires_grad)
        ctx.runner, ctx.nstart, ctx.ngrad, ctx.saved, ctx.steps = runner, nstart, ngrad, saved, steps
        ctx.x_needs_grad = x.requires_grad
        ctx.N, ctx.splits, ctx.out_shape = x.shape[0], splits, tuple(out.shape)
        outs, n0 = [], 0
        for n in splits:
            outs.append(out[n0:n0 + n].permute(0, 3, 1, 2))
            n0 += n
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        runner = ctx.runner
        # NHWC gradient of the leading ``ngrad`` images (the only ones the backward pass reads)
        parts, n0 = [], 0
        for n, g in zip(ctx.splits, gouts):
            if n0 >= ctx.ngrad:
                break
            take = min(n, ctx.ngrad - n0)
            if g is None:
                ref = next(t for t in gouts if t is not None)
                parts.append(torch.zeros((take,) + ctx.out_shape[1:], device=ref.device, dtype=torch.float32))
            else:
                parts.append(g.permute(0, 2, 3, 1)[:take])
            n0 += n
        gout = parts[0].contiguous() if len(parts) == 1 else torch.cat(parts, dim=0)
        dx, grads = runner.backward(gout, ctx.saved, ctx.nstart, ctx.ngrad, ctx.x_needs_grad, ctx.steps)
        ctx.saved = None
        if dx is not None and ctx.ngrad < ctx.N:
            full = torch.zeros((ctx.N,) + tuple(dx.shape[1:]), device=dx.device, dtype=dx.dtype)
            full[:ctx.ngrad] = dx
            dx = full
        return (None, None, None, None, None, dx) + tuple(grads.get(p) for p in runner.params)


def run_pass_groups(runner, x_nhwc, splits, nstart=None, ngrad=None, train=True):
    """Runs the stack over the NHWC batch and returns one logical-NCHW output view per entry of ``splits`` (image counts)."""
    n = x_nhwc.shape[0]
    nstart = tuple(nstart) if nstart is not None else (0, n)
    ngrad = n if ngrad is None else int(ngrad)
    if not torch.is_grad_enabled():
        ngrad = 0       # torch.no_grad(): ctx.needs_input_grad still reports the parameters, but no backward pass will come -- keep nothing
    splits = tuple(int(v) for v in splits)
    if sum(splits) != n:
        raise ValueError("splits %s do not add up to the batch size %d" % (splits, n))
    return list(_PassFn.apply(runner, nstart, ngrad, bool(train), splits, x_nhwc, *runner.params))


def run_pass(runner, x_nhwc, nstart=None, ngrad=None, train=True):
    """Single-output form: NHWC in, NHWC out."""
    out = run_pass_groups(runner, x_nhwc, (x_nhwc.shape[0],), nstart, ngrad, train)[0]
    return out.permute(0, 2, 3, 1)


# ---- layout helpers (zero-copy where the memory already is NHWC) -----------------------------------------------
def to_nhwc(t):
    """Logical NCHW tensor -> contiguous [N,H,W,C] view/copy."""
    if t.dim() != 4:
        raise ValueError("expected a 4-D NCHW tensor, got shape %s" % (tuple(t.shape),))
    if t.dtype != torch.float32:
        t = t.float()
    if t.shape[1] == 1:                              # single channel: NCHW memory IS NHWC memory
        return t.contiguous().reshape(t.shape[0], t.shape[2], t.shape[3], 1)
    return t.permute(0, 2, 3, 1).contiguous()       # no-op when t is channels_last


def to_nchw_view(t_nhwc):
    """[N,H,W,C] -> logical NCHW view (channels_last strides, no copy)."""
    return t_nhwc.permute(0, 3, 1, 2)
