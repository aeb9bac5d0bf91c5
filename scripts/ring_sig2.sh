#!/bin/bash
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 10 300 python3 -m pytest tests/test_gpu_kernels.py -x -q -k "wino" > $OUT/ring_tests.txt 2>&1 || { tail -30 $OUT/ring_tests.txt; exit 1; }
tail -2 $OUT/ring_tests.txt
SIGS="1 2 4 8" bash scripts/ring_sig.sh
