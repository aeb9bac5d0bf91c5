#!/bin/bash
# Kernel sequence of ONE RANK's replayed data-parallel step at 2 triplets (communicator of one) in both SyncBN exchange forms
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
for S in rccl p2p; do
  rm -rf /tmp/dpt_$S
  AESR_FORCE_DP=1 AESR_SYNCBN=$S rocprofv3 --kernel-trace -d /tmp/dpt_$S -o res -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 4 --triplets 2 --no-cpu-baseline --no-roofline --no-secondary > $OUT/r04_dp_${S}_bench.json 2> /dev/null
  DB=$(find /tmp/dpt_$S -name "*.db" | head -1)
  python3 $GRAFT_REPO_ROOT/scripts/kseq_last.py $DB > $OUT/r04_dp_${S}_kernel_sequence.txt
  python3 $GRAFT_REPO_ROOT/scripts/kstats_last.py $DB 4 > $OUT/r04_dp_${S}_kernel_stats.txt
  echo "== SyncBN exchange $S"; head -1 $OUT/r04_dp_${S}_kernel_stats.txt; grep -E "bn_|nccl|rccl|Reduce|p2p_tick|copy" $OUT/r04_dp_${S}_kernel_stats.txt | cut -c1-150
done
