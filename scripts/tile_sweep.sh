#!/bin/bash
# which tile does the planner pick, and how do alternatives perform (32->32 @162, N=36 forward)
AESR_PLAN_DEBUG=1 python scripts/bench_one.py fwd 36 162 162 32 32 1 2>&1 | grep "aesr plan" | head -3
for t in "1,18,27" "1,27,18" "1,14,18" "1,18,18" "1,9,54" "1,16,32" "1,18,14" "1,27,9" "1,9,27"; do
  AESR_IGEMM_TILE=$t python - <<PY
import subprocess,sys,os,torch
sys.path.insert(0,'.')
from superresolution_aniso_mri_amd import _hip as hip
L=hip.lib
N,H,W,Cin,Cout=36,162,162,32,32
x=torch.randn(N,H,W,Cin,device='cuda'); w=torch.randn(Cout,Cin,3,3,device='cuda')*0.05; b=torch.zeros(Cout,device='cuda')
pf=torch.empty(L.aesr_conv2d_packed_floats(Cout,Cin,3,0),device='cuda')
hip.check(L.aesr_conv2d_pack(hip.ptr(w),hip.ptr(pf),Cout,Cin,3,0,hip.stream()),'p')
out=torch.empty(N,H,W,Cout,device='cuda')
f=lambda: hip.check(L.aesr_conv2d_fwd(hip.ptr(x),hip.ptr(pf),hip.ptr(b),hip.ptr(out),N,H,W,Cin,Cout,3,1,1,0.01,hip.stream()),'f')
f(); torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): f()
e1.record(); torch.cuda.synchronize()
t=e0.elapsed_time(e1)/10*1e-3
print("tile $t: %.1f us  %.1f TF" % (t*1e6, 2.0*N*H*W*Cin*Cout*9/t/1e12))
PY
done
