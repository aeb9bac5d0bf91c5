#!/bin/bash
# expand tile height of the thin kernels (THIN_ETH) against the number of workgroup rounds: the grid is N x ceil(H / ETH) x ceil(W / 32) workgroups of
# 256 threads, 8 resident per CU = 2 048 at a time -- 36 x 11 x 6 = 2 376 at ETH = 16 is 1.16 rounds (the second one 16 % full)
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_thin_eth.txt
: > $OUT
C=$GRAFT_REPO_ROOT/superresolution_aniso_mri_amd/csrc
L=$GRAFT_REPO_ROOT/superresolution_aniso_mri_amd/libaesr_hip.so
cp $L /tmp/libaesr_orig.so
for E in 16 18 20 27 32 12 9; do
  /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wall -Wno-unused-result -ffp-contract=off -DTHIN_ETH=$E -c $C/conv_thin.hip -o /tmp/conv_thin_$E.o
  OBJS=$(ls $C/build/*.o | grep -v conv_thin.o)
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $L $OBJS /tmp/conv_thin_$E.o -ldl
  echo "THIN_ETH=$E" >> $OUT
  python3 $GRAFT_REPO_ROOT/scripts/r04_thin_eth.py 2>/dev/null >> $OUT
done
cp /tmp/libaesr_orig.so $L
cat $OUT
