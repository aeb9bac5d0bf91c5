#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 10 400 python3 -m pytest tests/test_gpu_kernels.py -x -q -k "wino" 2>&1 | tail -2
for N in 3 6 18; do for M in 0 2; do
AESR_BENCH_N=$N AESR_WINO_RING=$M AESR_PLAN_DEBUG=1 timeout -k 10 200 python3 scripts/bench_wino.py all > $OUT/ring_small_n${N}_m${M}.txt 2>&1
done; done
