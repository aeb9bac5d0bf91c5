#!/bin/bash
# whole-step A/B on one box: the ring kernel off / by cost / always, C2 and C3
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 10 400 python3 -m pytest tests/test_gpu_kernels.py -x -q -k "wino" > $OUT/ring_tests.txt 2>&1 || { tail -30 $OUT/ring_tests.txt; exit 1; }
tail -2 $OUT/ring_tests.txt
for MODE in 0 1 2; do for CFG in c2 c3; do
AESR_WINO_RING=$MODE python3 bench.py --steps 30 --warmup 10 --config $CFG --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ring mode $MODE $CFG: %.3f ms/step  loss %.6f' % (d['ms_per_step'], d['final_loss']))"
done; done
