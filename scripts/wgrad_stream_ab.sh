#!/bin/bash
# weight gradients on a side stream (AESR_WGRAD_STREAM) off / on: replayed-graph step at each shard size, C2 and C3, on one box
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out
TAG=${1:-r03_wgrad_stream_ab}
timeout -k 10 300 python3 -m pytest tests/test_gpu_step.py tests/test_gpu_dp.py -q -x 2>&1 | tail -3
: > $OUT/$TAG.txt
for CFG in c2 c3; do for T in 1 2 3 6 12; do for M in 0 1; do
AESR_WGRAD_STREAM=$M python3 bench.py --steps 40 --warmup 10 --config $CFG --triplets $T --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null \
  | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$CFG triplets $T wgrad stream $M: %.3f ms/step  loss %.6f' % (d['ms_per_step'], d['final_loss']))" >> $OUT/$TAG.txt
done; done; done
cat $OUT/$TAG.txt
