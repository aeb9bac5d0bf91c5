#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 10 400 python3 -m pytest tests/test_gpu_kernels.py -x -q -k "wino" > $OUT/ksplit_tests.txt 2>&1 || { tail -40 $OUT/ksplit_tests.txt | cut -c1-300; exit 1; }
tail -2 $OUT/ksplit_tests.txt
: > $OUT/r03_ksplit_layers.txt
for N in 3 6 36; do
  echo "== images N=$N (VGG: 2N/3), forward" >> $OUT/r03_ksplit_layers.txt
  AESR_BENCH_N=$N timeout -k 10 300 python3 scripts/bench_ksplit.py all >> $OUT/r03_ksplit_layers.txt 2>&1
  echo "== images N=$N, data gradient" >> $OUT/r03_ksplit_layers.txt
  AESR_BENCH_N=$N timeout -k 10 300 python3 scripts/bench_ksplit.py all --dgrad >> $OUT/r03_ksplit_layers.txt 2>&1
done
cut -c1-220 $OUT/r03_ksplit_layers.txt
