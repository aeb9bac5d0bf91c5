#!/bin/bash
# what a rank of a data-parallel run executes per step, measured on one GPU with a communicator of one (AESR_FORCE_DP=1): the SyncBN
# kernels and the 9 collectives are in the step; both graph forms, next to the plain single-process step
OUT=$GRAFT_REPO_ROOT/gpurun_out
TAG=${1:-r03_dp_one_rank}
: > $OUT/$TAG.txt
for CFG in c2 c3; do for T in 1 2; do
for MODE in none segments whole; do
  if [ $MODE = none ]; then unset AESR_FORCE_DP AESR_DP_GRAPH; else export AESR_FORCE_DP=1 AESR_DP_GRAPH=$MODE; fi
  python3 bench.py --steps 40 --warmup 10 --config $CFG --triplets $T --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null \
   | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$CFG triplets $T dp $MODE: %.3f ms/step  loss %.6f  (%s)' % (d['ms_per_step'], d['final_loss'], d['config']['launch']))" >> $OUT/$TAG.txt
done; done; done
unset AESR_FORCE_DP AESR_DP_GRAPH
cat $OUT/$TAG.txt
