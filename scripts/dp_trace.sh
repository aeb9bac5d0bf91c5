#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for MODE in none whole; do
  if [ $MODE = none ]; then unset AESR_FORCE_DP AESR_DP_GRAPH; else export AESR_FORCE_DP=1 AESR_DP_GRAPH=$MODE; fi
  rm -rf /tmp/dpt_$MODE
  rocprofv3 --kernel-trace -d /tmp/dpt_$MODE -o res -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 4 --config c3 --triplets 2 --no-cpu-baseline --no-roofline --no-secondary > /dev/null 2>&1
  DB=$(find /tmp/dpt_$MODE -name "*.db" | head -1)
  python3 $GRAFT_REPO_ROOT/scripts/kstats.py $DB 12 > $OUT/dpt_${MODE}_stats.txt
  python3 $GRAFT_REPO_ROOT/scripts/kseq_last.py $DB > $OUT/dpt_${MODE}_seq.txt
  tail -1 $OUT/dpt_${MODE}_seq.txt; head -1 $OUT/dpt_${MODE}_stats.txt
done
