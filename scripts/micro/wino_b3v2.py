#!/usr/bin/env python
"""Driver of the stand-alone two-waves-per-SIMD bf16 x 3 resident-filter kernel (scripts/micro/wino_b3v2.hip): correctness against fp64 on the CPU
(small and ragged cases, every epilogue form) and against the library's f32-MFMA kernel, then launch times of both on the 32-channel layers of BASELINE
configuration 2 (32 -> 32 @ 160 x 160, N = 4 / 12 / 36) under HIP events.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -Isuperresolution_aniso_mri_amd/csrc -shared -fPIC \
          scripts/micro/wino_b3v2.hip -o scripts/micro/libwino_b3v2.so
    python scripts/micro/wino_b3v2.py [grid]"""
import ctypes
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from superresolution_aniso_mri_amd import _hip as hip  # noqa: E402
from scripts.bench_wino import pack_wino  # noqa: E402

P = ctypes.CDLL(os.environ.get("B3_LIB") or os.path.join(ROOT, "scripts", "micro", "libwino_b3v2.so"))
P.b3v2_conv.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 6 + [ctypes.c_float] + [ctypes.c_int] * 4 + [ctypes.c_void_p]
GRID = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ACT_NONE, ACT_LRELU, ACT_RELU, ACT_SIGMOID = hip.ACT_NONE, hip.ACT_LRELU, hip.ACT_RELU, hip.ACT_SIGMOID


def pack_u3(w):
    """w [Cout, Cin, 3, 3] fp32 on the GPU -> the kernel's filter image: U = G g G^T (fp64 -> fp32), split x = hi + mid + lo (truncation, exact),
    [chunk][cout tile][position 16][cout block 2][plane 3][lane 64][4 bf16]: lane (l15, g) = cout 32 tile + 16 block + l15, channels 16 chunk + 4 g .. + 3."""
    Cout, Cin = w.shape[:2]
    CinP, CoutP = (Cin + 15) // 16 * 16, (Cout + 31) // 32 * 32
    G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64, device=w.device)
    U = torch.zeros(16, CoutP, CinP, device=w.device)
    U[:, :Cout, :Cin] = torch.einsum("ak,ockl,bl->aboc", G, w.double(), G).reshape(16, Cout, Cin).float()
    hi = U.view(torch.int32) & -65536
    r = U - hi.view(torch.float32)
    mid = r.view(torch.int32) & -65536
    r2 = r - mid.view(torch.float32)
    lo = r2.view(torch.int32)
    assert int((lo & 0xffff).abs().max()) == 0, "the third term is not a bf16"
    p16 = ((torch.stack([hi, mid, lo], 0) >> 16) & 0xffff)                  # [plane][xi][co][ci]
    p16 = p16.reshape(3, 16, CoutP // 32, 2, 16, CinP // 16, 4, 4)           # plane, xi, cot, nb, l15, chunk, g, j
    out = p16.permute(5, 2, 1, 3, 0, 6, 4, 7).contiguous()                   # chunk, cot, xi, nb, plane, g, l15, j  (lane = 16 g + l15)
    return out.to(torch.int16).contiguous()


def run_b3(x, u3, b, Cout, act=ACT_LRELU, slope=0.01, ysave=None, mask_act=ACT_NONE, in_up2=0, out_sum2=0):
    N, Hs, Ws, Cin = x.shape
    H, W = (Hs * 2, Ws * 2) if in_up2 else (Hs, Ws)
    out = torch.full((N, H // 2, W // 2, Cout) if out_sum2 else (N, H, W, Cout), float("nan"), device="cuda")
    rc = P.b3v2_conv(x.data_ptr(), u3.data_ptr(), b.data_ptr() if b is not None else None, ysave.data_ptr() if ysave is not None else None, out.data_ptr(),
                     N, H, W, Cin, Cout, act, slope, mask_act, in_up2, out_sum2, GRID, None)
    assert rc == 0, rc
    return out


def ref64(x, w, b, act, slope, ysave=None, mask_act=ACT_NONE, in_up2=0, out_sum2=0):
    xx = x.cpu().double().permute(0, 3, 1, 2)
    if in_up2:
        xx = F.interpolate(xx, scale_factor=2, mode="nearest")
    y = F.conv2d(xx, w.cpu().double(), b.cpu().double() if b is not None else None, padding=1)
    if out_sum2:
        return (F.avg_pool2d(y, 2) * 4).permute(0, 2, 3, 1)
    if act == ACT_LRELU:
        y = F.leaky_relu(y, slope)
    elif act == ACT_RELU:
        y = F.relu(y)
    elif act == ACT_SIGMOID:
        y = torch.sigmoid(y)
    if ysave is not None:
        ys = ysave.cpu().double().permute(0, 3, 1, 2)
        d = torch.ones_like(ys)
        if mask_act == ACT_LRELU:
            d = torch.where(ys > 0, 1.0, slope)
        elif mask_act == ACT_RELU:
            d = (ys > 0).double()
        y = y * d
    return y.permute(0, 2, 3, 1)


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


torch.manual_seed(0)
print("# bf16 x 3 resident-filter kernel, two waves per SIMD (scripts/micro/wino_b3v2.hip), grid %d" % GRID)
worst = 0.0
cases = [
    # N, H, W, Cin, Cout, act, mask_act, in_up2, out_sum2
    (2, 17, 23, 32, 32, ACT_LRELU, None, 0, 0),
    (3, 40, 40, 32, 32, ACT_LRELU, None, 0, 0),
    (1, 8, 8, 32, 32, ACT_NONE, None, 0, 0),
    (2, 33, 9, 16, 32, ACT_RELU, None, 0, 0),
    (2, 20, 28, 32, 64, ACT_SIGMOID, None, 0, 0),
    (2, 21, 19, 20, 48, ACT_LRELU, None, 0, 0),
    (2, 24, 24, 32, 32, ACT_NONE, ACT_LRELU, 0, 0),
    (2, 19, 27, 32, 32, ACT_NONE, ACT_RELU, 0, 0),
    (2, 24, 32, 32, 32, ACT_LRELU, None, 1, 0),
    (2, 24, 32, 32, 32, ACT_NONE, None, 0, 1),
    (5, 64, 64, 32, 32, ACT_LRELU, None, 0, 0),
]
for (N, H, W, Cin, Cout, act, mask_act, up2, sum2) in ([] if os.environ.get("B3_TIME_ONLY") else cases):
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.06
    b = torch.randn(Cout, device="cuda") * 0.1
    x = torch.randn(N, H // 2 if up2 else H, W // 2 if up2 else W, Cin, device="cuda")
    ys = torch.randn(N, H, W, Cout, device="cuda") if mask_act is not None else None
    u3 = pack_u3(w)
    y = run_b3(x, u3, b, Cout, act, 0.01, ys, mask_act if mask_act is not None else ACT_NONE, up2, sum2)
    torch.cuda.synchronize()
    ref = ref64(x, w, b, act, 0.01, ys, mask_act if mask_act is not None else ACT_NONE, up2, sum2)
    e = rel(y.cpu(), ref)
    worst = max(worst, e if e == e else 1.0)
    print("  N %d, %3d x %3d, %2d -> %2d, act %d, mask %s, up2 %d, sum2 %d: rel-L2 against fp64 %.2e, max abs %.2e%s" % (
        N, H, W, Cin, Cout, act, mask_act, up2, sum2, e, float((y.cpu().double() - ref).abs().max()), "" if e < 1e-6 else "   <-- WRONG"))
print("  worst rel-L2 %.2e  (%s)" % (worst, "PASS" if worst < 1e-6 else "FAIL"))
if worst >= 1e-6 and not os.environ.get("B3_TIME_ONLY"):
    sys.exit(1)

w = torch.randn(32, 32, 3, 3, device="cuda") * 0.06
b = torch.randn(32, device="cuda") * 0.1
u3, uf = pack_u3(w), pack_wino(w, 32, 32, 0)


def run_lib(x, out):
    N, H, W, _ = x.shape
    hip.check(hip.lib.aesr_conv2d_wino_fwd(hip.ptr(x), hip.ptr(uf), hip.ptr(b), hip.ptr(out), N, H, W, 32, 32, 1, 0.01, hip.stream()), "fwd")


for N in (4, 12, 36):
    x = torch.randn(N, 160, 160, 32, device="cuda")
    o1, o2 = torch.empty(N, 160, 160, 32, device="cuda"), torch.empty(N, 160, 160, 32, device="cuda")

    def run_p():
        rc = P.b3v2_conv(x.data_ptr(), u3.data_ptr(), b.data_ptr(), None, o1.data_ptr(), N, 160, 160, 32, 32, ACT_LRELU, 0.01, ACT_NONE, 0, 0, GRID, None)
        assert rc == 0

    run_p()
    run_lib(x, o2)
    torch.cuda.synchronize()
    same_twice = True
    ref = o1.clone()
    for _ in range(5):
        run_p()
        torch.cuda.synchronize()
        same_twice &= torch.equal(ref, o1)
    tp, tl = timeit(run_p), timeit(lambda: run_lib(x, o2))
    print("  %2d x 160 x 160, 32 -> 32: b3 vs library rel-L2 %.2e, 6 runs bitwise equal: %s | b3 %.1f us, conv_wino_res_f32 %.1f us, ratio %.2f" % (
        N, rel(o1, o2), same_twice, tp, tl, tl / tp))
