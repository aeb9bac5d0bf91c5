#!/bin/bash
# A/B of the expand tile height of the thin kernels (THIN_ETH): rebuilds conv_thin.o with another value into a scratch copy of the library
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_thin_ab.txt
: > $OUT
C=$GRAFT_REPO_ROOT/superresolution_aniso_mri_amd/csrc
cp $GRAFT_REPO_ROOT/superresolution_aniso_mri_amd/libaesr_hip.so /tmp/libaesr_orig.so
for E in 16 32 8; do
  /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wall -Wno-unused-result -ffp-contract=off -DTHIN_ETH=$E -c $C/conv_thin.hip -o /tmp/conv_thin_$E.o
  OBJS=$(ls $C/build/*.o | grep -v conv_thin.o)
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $GRAFT_REPO_ROOT/superresolution_aniso_mri_amd/libaesr_hip.so $OBJS /tmp/conv_thin_$E.o -ldl
  echo "THIN_ETH=$E" >> $OUT
  python3 $GRAFT_REPO_ROOT/scripts/bench_small.py "" 2>/dev/null | grep -E "stemconv|cout1" >> $OUT
done
cp /tmp/libaesr_orig.so $GRAFT_REPO_ROOT/superresolution_aniso_mri_amd/libaesr_hip.so
cat $OUT
