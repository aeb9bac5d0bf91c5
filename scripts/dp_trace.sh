#!/bin/bash
# Kernel statistics + sequence of ONE RANK's replayed data-parallel step (communicator of one) in both SyncBN exchange forms: dp_trace.sh [tag] [cfg] [triplets]
TAG=${1:-r05}; CFG=${2:-c2}; T=${3:-2}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
for S in rccl p2p; do
  rm -rf /tmp/dpt_$S
  AESR_FORCE_DP=1 AESR_SYNCBN=$S rocprofv3 --kernel-trace -d /tmp/dpt_$S -o res -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 4 --config $CFG --triplets $T --no-cpu-baseline --no-roofline --no-secondary > $OUT/${TAG}_dp_${CFG}_${S}_bench.json 2> /dev/null
  DB=$(find /tmp/dpt_$S -name "*.db" | head -1)
  python3 $GRAFT_REPO_ROOT/scripts/kseq_last.py $DB > $OUT/${TAG}_dp_${CFG}_${S}_kernel_sequence.txt
  python3 $GRAFT_REPO_ROOT/scripts/kstats_last.py $DB 4 > $OUT/${TAG}_dp_${CFG}_${S}_kernel_stats.txt
  echo "== SyncBN exchange $S"; head -1 $OUT/${TAG}_dp_${CFG}_${S}_kernel_stats.txt; grep -E "bn_|nccl|rccl|Reduce|p2p_tick|copy" $OUT/${TAG}_dp_${CFG}_${S}_kernel_stats.txt | cut -c1-150
done
