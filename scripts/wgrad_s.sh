#!/bin/bash
# time the wgrad of one layer for several split counts: wgrad_s.sh N H W Cin Cout "S S ..."
for s in $6; do
  echo -n "S=$s: "
  AESR_WGRAD_S=$s python - "$1" "$2" "$3" "$4" "$5" <<'PY'
import sys, torch
sys.path.insert(0, "/root/repo")
from superresolution_aniso_mri_amd import _hip as hip
L = hip.lib
N, H, W, Cin, Cout = [int(v) for v in sys.argv[1:6]]
x = torch.randn(N, H, W, Cin, device="cuda"); dy = torch.randn(N, H, W, Cout, device="cuda")
dw = torch.empty(Cout, Cin, 3, 3, device="cuda"); db = torch.empty(Cout, device="cuda")
ws = torch.empty(L.aesr_conv2d_wgrad_workspace_floats(N, H, W, Cin, Cout, 3, 1), device="cuda")
f = lambda: hip.check(L.aesr_conv2d_wgrad(hip.ptr(x), hip.ptr(dy), hip.ptr(dw), hip.ptr(db), hip.ptr(ws), N, H, W, Cin, Cout, 3, 1, hip.stream()), "w")
for _ in range(3): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): f()
e1.record(); torch.cuda.synchronize()
print("%.1f us" % (e0.elapsed_time(e1) * 100))
PY
done
