#!/bin/bash
# rocprofv3 kernel statistics of the bench step at a small shard: profile_small.sh <tag> <triplets>
set -e
TAG=$1; T=${2:-2}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace -d /tmp/prof_$TAG -o res -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 0 --triplets $T --no-cpu-baseline --no-roofline --no-secondary > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
DB=$(find /tmp/prof_$TAG -name "*.db" | head -1)
python3 $GRAFT_REPO_ROOT/scripts/kstats.py $DB 8 > $OUT/${TAG}_kernel_stats.txt
head -70 $OUT/${TAG}_kernel_stats.txt
