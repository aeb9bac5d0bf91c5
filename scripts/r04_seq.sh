#!/bin/bash
# Kernel sequence of the REPLAYED (captured graph) step at a small shard: r04_seq.sh <tag> <triplets> [config]
#   -> gpurun_out/<tag>_kernel_sequence.txt (last step: start offset, duration, gap to the previous kernel), <tag>_kernel_stats.txt
set -e
TAG=$1; T=${2:-2}; CFG=${3:-c2}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
rm -rf /tmp/seq_$TAG
rocprofv3 --kernel-trace -d /tmp/seq_$TAG -o res -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 4 --triplets $T --config $CFG --no-cpu-baseline --no-roofline --no-secondary > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
DB=$(find /tmp/seq_$TAG -name "*.db" | head -1)
python3 $GRAFT_REPO_ROOT/scripts/kseq_last.py $DB > $OUT/${TAG}_kernel_sequence.txt
tail -3 $OUT/${TAG}_kernel_sequence.txt
cat $OUT/${TAG}_bench.json | cut -c1-400
