#!/bin/bash
# experiment: where in a chunk the ring kernel signals the arrival of the next one (RG_SIG_POS), 8-wave workgroups everywhere
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT/superresolution_aniso_mri_amd/csrc
for SIG in ${SIGS:-4 8 12 15}; do
  /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wall -Wno-unused-result -ffp-contract=off -fno-slp-vectorize -DRG_SIG_POS=$SIG -c conv_wino_ring.hip -o build/conv_wino_ring.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libaesr_hip.so build/*.o -ldl
  (cd $GRAFT_REPO_ROOT && AESR_RING_SHAPE=${SHAPE:-0,0,0,8} timeout -k 10 200 python3 scripts/bench_wino.py vgg > $OUT/ring_sig${SIG}_vgg.txt 2>&1)
  echo "SIG=$SIG"; grep -E "GF \||TOTAL" $OUT/ring_sig${SIG}_vgg.txt | awk '{print $7, $8, $9, "|", $(NF-6), $(NF-5)}' | tr '\n' ';'; echo
done
