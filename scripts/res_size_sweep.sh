#!/bin/bash
# Does conv_wino_res_f32 run faster per round of items when its input and output stay in the Infinity Cache?  32 -> 32 @ 162 x 162, N images,
# 32-cout workgroups forced (one item = 256 MFMAs per wave), 10 back-to-back launches on the same buffers (time_one.py): rounds = N x 441 / 2 048.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r05_res_size_sweep}.txt
: > $OUT
for N in 4 9 14 18 23 27 36 54 72; do
  T=$(AESR_WINO_RES_TN=32 timeout -k 10 100 python3 $R/scripts/time_one.py fwd $N 162 162 32 32 2>/dev/null | tail -n 1)
  python3 -c "
N=$N; t=$T; items=N*441; rounds=items/2048.0; import math
print('N=%3d  in+out %6.1f MB  %7.1f us  items %6d = %5.2f rounds (ceil %d)  %.2f us per round (ceil)  %.1f TF executed' % (N, 2*N*162*162*32*4/1e6, t, items, rounds, math.ceil(rounds), t/math.ceil(rounds), 2.0*N*162*162*32*32*9/2.25/t/1e6))" | tee -a $OUT
done
