#!/bin/bash
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 10 400 python3 -m pytest tests/test_gpu_kernels.py -x -q -k "wino" > $OUT/xcd_tests.txt 2>&1 || { tail -30 $OUT/xcd_tests.txt; exit 1; }
tail -2 $OUT/xcd_tests.txt
for X in 0 1 0 1; do
AESR_WINO_XCD=$X python3 bench.py --steps 40 --warmup 10 --config c2 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('xcd map $X c2: %.3f ms/step' % d['ms_per_step'])"
done
AESR_WINO_XCD=0 bash scripts/profile_all.sh r03_xcd0 c2 > $OUT/xcd0_prof.txt 2>&1
AESR_WINO_XCD=1 bash scripts/profile_all.sh r03_xcd1 c2 > $OUT/xcd1_prof.txt 2>&1
