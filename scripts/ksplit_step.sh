#!/bin/bash
# whole step with and without the channel split (AESR_RING_KSPLIT=1 forces unsplit plans), every shard size, one box
OUT=$GRAFT_REPO_ROOT/gpurun_out
: > $OUT/r03_ksplit_step.txt
for CFG in c3 c2; do for T in 1 2 3 6 12; do for M in 1 0; do
if [ $M = 1 ]; then export AESR_RING_KSPLIT=1; else unset AESR_RING_KSPLIT; fi
python3 bench.py --steps 40 --warmup 10 --config $CFG --triplets $T --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null \
  | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$CFG triplets $T channel split %s: %.3f ms/step  loss %.6f' % ('off' if $M else 'planned', d['ms_per_step'], d['final_loss']))" >> $OUT/r03_ksplit_step.txt
done; done; done
unset AESR_RING_KSPLIT
cat $OUT/r03_ksplit_step.txt
timeout -k 10 1000 python3 -m pytest tests -m gpu -q -x > $OUT/r03_gpu_tests.txt 2>&1; tail -5 $OUT/r03_gpu_tests.txt | cut -c1-250
