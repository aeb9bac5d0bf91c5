#!/usr/bin/env python
"""Driver of the bf16 x 3 resident-filter PROTOTYPE (scripts/micro/wino_b3_proto.hip): correctness against fp64 on the CPU (small case) and against
the library's f32-MFMA kernel (full size), then launch times of both on 32 -> 32 @ 162 x 162 x {4, 12, 36} images.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -shared -fPIC scripts/micro/wino_b3_proto.hip -o scripts/micro/libwino_b3_proto.so
    python scripts/micro/wino_b3_proto.py [grid]"""
import ctypes
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from superresolution_aniso_mri_amd import _hip as hip  # noqa: E402
from scripts.bench_wino import pack_wino  # noqa: E402

P = ctypes.CDLL(os.environ.get("B3_LIB") or os.path.join(ROOT, "scripts", "micro", "libwino_b3_proto.so"))
P.b3_conv_fwd.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 3 + [ctypes.c_float, ctypes.c_int, ctypes.c_void_p]
GRID = int(sys.argv[1]) if len(sys.argv) > 1 else 256


def pack_u3(w):
    """w [32, 32, 3, 3] fp32 on the GPU -> the prototype's filter image: U = G g G^T (fp64 -> fp32), split x = hi + mid + lo (truncation, exact),
    [position 16][cout block 2][plane 3][lane 64][8 bf16]: lane L holds cout 16 nb + (L & 15), channels 8 (L >> 4) .. + 7."""
    G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64, device=w.device)
    U = torch.einsum("ak,ockl,bl->aboc", G, w.double(), G).reshape(16, 32, 32).float().contiguous()       # [xi][co][ci]
    bits = U.view(torch.int32)
    hi = bits & -65536
    r = U - hi.view(torch.float32)
    mid = r.view(torch.int32) & -65536
    r2 = r - mid.view(torch.float32)
    lo = r2.view(torch.int32)
    assert int((lo & 0xffff).abs().max()) == 0, "the third term is not a bf16"
    planes = torch.stack([hi, mid, lo], 0)                                   # [plane][xi][co][ci] int32 with the bf16 in the upper half
    p16 = ((planes >> 16) & 0xffff).to(torch.int32)
    out = torch.empty((16, 2, 3, 64, 8), dtype=torch.int32, device=w.device)
    L = torch.arange(64, device=w.device)
    m, g = L & 15, L >> 4
    for nb in range(2):
        co = nb * 16 + m                                                     # [64]
        for j in range(8):
            ci = 8 * g + j
            out[:, nb, :, :, j] = p16[:, :, co, ci].permute(1, 0, 2)          # [xi][plane][lane]
    return out.to(torch.int16).contiguous()


def run_proto(x, u3, b, slope=0.01):
    N, H, W, _ = x.shape
    out = torch.empty(N, H, W, 32, device="cuda")
    rc = P.b3_conv_fwd(x.data_ptr(), u3.data_ptr(), b.data_ptr(), out.data_ptr(), N, H, W, slope, GRID, None)
    assert rc == 0, rc
    return out


def run_lib(x, uf, b, slope=0.01):
    N, H, W, _ = x.shape
    out = torch.empty(N, H, W, 32, device="cuda")
    hip.check(hip.lib.aesr_conv2d_wino_fwd(hip.ptr(x), hip.ptr(uf), hip.ptr(b), hip.ptr(out), N, H, W, 32, 32, 1, slope, hip.stream()), "fwd")
    return out


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def timeit(fn, n=20):
    for _ in range(3):
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
w = torch.randn(32, 32, 3, 3, device="cuda") * 0.06
b = torch.randn(32, device="cuda") * 0.1
u3, uf = pack_u3(w), pack_wino(w, 32, 32, 0)
print("# bf16 x 3 resident-filter prototype (scripts/micro/wino_b3_proto.hip), 32 -> 32 channels, LeakyReLU(0.01), grid %d" % GRID)
for (N, H, W) in [(2, 17, 23), (3, 40, 40)]:
    x = torch.randn(N, H, W, 32, device="cuda")
    ref = F.leaky_relu(F.conv2d(x.cpu().double().permute(0, 3, 1, 2), w.cpu().double(), b.cpu().double(), padding=1), 0.01).permute(0, 2, 3, 1)
    yp, yl = run_proto(x, u3, b), run_lib(x, uf, b)
    torch.cuda.synchronize()
    print("  %d x %d x %d: rel-L2 against fp64: prototype %.2e, library f32 kernel %.2e; max |prototype - fp64| %.2e" % (
        N, H, W, rel(yp.cpu(), ref), rel(yl.cpu(), ref), float((yp.cpu().double() - ref).abs().max())))
for N in (4, 12, 36):
    x = torch.randn(N, 162, 162, 32, device="cuda")
    yp, yl = run_proto(x, u3, b), run_lib(x, uf, b)
    torch.cuda.synchronize()
    tp, tl = timeit(lambda: run_proto(x, u3, b)), timeit(lambda: run_lib(x, uf, b))
    print("  %2d x 162 x 162: prototype vs library output rel-L2 %.2e | prototype %.1f us, conv_wino_res_f32 %.1f us (incl. torch.empty + ctypes call), ratio %.2f" % (
        N, rel(yp, yl), tp, tl, tl / tp))
