#!/bin/bash
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 10 300 python3 -m pytest tests/test_gpu_kernels.py -x -q -k "wino" > $OUT/ring_tests.txt 2>&1 || { tail -30 $OUT/ring_tests.txt; exit 1; }
tail -2 $OUT/ring_tests.txt
for F in 0 2 4 6 8; do
AESR_WINO_FLAGS=$F timeout -k 10 200 python3 scripts/bench_wino.py vgg > $OUT/ring_f${F}_vgg.txt 2>&1
AESR_WINO_FLAGS=$F timeout -k 10 200 python3 scripts/bench_wino.py ae > $OUT/ring_f${F}_ae.txt 2>&1
done
