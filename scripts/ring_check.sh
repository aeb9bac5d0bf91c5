#!/bin/bash
# ring kernel: parity tests, then the VGG / AE layer tables with and without it
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 10 300 python3 -m pytest tests/test_gpu_kernels.py -x -q -k "wino" > $OUT/ring_tests.txt 2>&1 || { tail -30 $OUT/ring_tests.txt; exit 1; }
tail -3 $OUT/ring_tests.txt
AESR_PLAN_DEBUG=1 timeout -k 10 200 python3 scripts/bench_wino.py vgg > $OUT/ring_vgg.txt 2>&1
grep -E "TOTAL|plan\] ring" $OUT/ring_vgg.txt | sort | uniq | tail -30
AESR_PLAN_DEBUG=1 timeout -k 10 200 python3 scripts/bench_wino.py ae > $OUT/ring_ae.txt 2>&1
grep -E "TOTAL" $OUT/ring_ae.txt
