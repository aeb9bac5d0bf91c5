#!/bin/bash
# where a small-shard weight-gradient launch spends its time: per-layer times at N = 6 images and the kernel's phase stamps
OUT=$GRAFT_REPO_ROOT/gpurun_out
AESR_BENCH_N=6 python3 scripts/bench_conv.py ae > $OUT/wgrad_small_n6.txt 2>&1
AESR_BENCH_N=6 AESR_WGRAD_WINO_DBG=1 AESR_PLAN_DEBUG=1 python3 scripts/bench_conv.py ae 2>&1 | grep -E "stamps|plan_wgrad" | sort | uniq -c | sort -rn | head -40 > $OUT/wgrad_small_n6_stamps.txt
cat $OUT/wgrad_small_n6.txt; cut -c1-330 $OUT/wgrad_small_n6_stamps.txt
