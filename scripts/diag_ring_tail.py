import os, sys
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superresolution_aniso_mri_amd import _hip as hip
from scripts.bench_wino import pack_wino
L = hip.lib
os.environ["AESR_WINO_RING"] = "2"
g = torch.Generator().manual_seed(5)
Cin, Cout, H, W = 128, 64, 18, 22
w = torch.randn(Cout, Cin, 3, 3, generator=g) / np.sqrt(Cin * 9)
b = torch.randn(Cout, generator=g)
up = pack_wino(w.cuda(), Cout, Cin, 0)
for N, shape in ((4, "1,4,4"), (5, "1,4,4"), (6, "1,3,5"), (40, "1,4,4"), (7, "1,2,2")):
    os.environ["AESR_RING_SHAPE"] = shape
    x = torch.randn(N, Cin, H, W, generator=g)
    ref = F.leaky_relu(F.conv2d(x.double(), w.double(), b.double(), padding=1), 0.01).permute(0, 2, 3, 1)
    out = torch.full((N, H, W, Cout), float("nan"), device="cuda")
    hip.check(L.aesr_conv2d_wino_fwd(hip.ptr(x.permute(0, 2, 3, 1).contiguous().cuda()), hip.ptr(up), hip.ptr(b.cuda()), hip.ptr(out), N, H, W, Cin, Cout, 1, 0.01, hip.stream()), "fwd")
    torch.cuda.synchronize()
    d = (out.cpu().double() - ref).abs()
    bad = (~(d < 1e-4)).nonzero()
    print("N=%d shape %s: bad %d of %d, nan %d; images %s rows %s cols %s chans %s; timeouts %d" % (
        N, shape, len(bad), d.numel(), int(torch.isnan(out).sum()), *[sorted(set(bad[:, k].tolist()))[:24] for k in range(4)], L.aesr_conv2d_wino_ring_timeouts()), flush=True)
