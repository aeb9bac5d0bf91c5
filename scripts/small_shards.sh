#!/bin/bash
# Replayed-graph step time of the bench workload at small shards (what strong scaling over 12 triplets runs on): small_shards.sh <tag> [config]
#   -> gpurun_out/<tag>_small_shards.txt ; plus the kernel sequence of one step at 2 triplets (rocprofv3 --kernel-trace)
set -e
TAG=$1; CFG=${2:-c2}
OUT=$GRAFT_REPO_ROOT/gpurun_out
: > $OUT/${TAG}_small_shards.txt
for T in 1 2 3 6 12; do
  python3 $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --config $CFG --triplets $T --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null \
    | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$CFG triplets $T: %.3f ms/step  %.1f slices/s' % (d['ms_per_step'], d['value']))" >> $OUT/${TAG}_small_shards.txt
done
cat $OUT/${TAG}_small_shards.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ss_$TAG
rocprofv3 --kernel-trace -d /tmp/ss_$TAG -o res -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 4 --config $CFG --triplets 2 --no-cpu-baseline --no-roofline --no-secondary > /dev/null 2>&1
DB=$(find /tmp/ss_$TAG -name "*.db" | head -1)
python3 $GRAFT_REPO_ROOT/scripts/kstats.py $DB 12 > $OUT/${TAG}_${CFG}_2triplets_kernel_stats.txt
python3 $GRAFT_REPO_ROOT/scripts/kseq.py $DB 12 > $OUT/${TAG}_${CFG}_2triplets_kernel_sequence.txt
tail -3 $OUT/${TAG}_${CFG}_2triplets_kernel_sequence.txt
