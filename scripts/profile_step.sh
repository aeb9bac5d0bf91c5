#!/bin/bash
# rocprofv3 kernel statistics of the bench step: profile_step.sh <tag> [config]   -> gpurun_out/<tag>_kernel_stats.txt
# (bench.py runs warmup + steps + 5 instrumented steps; kstats divides by the total number of steps executed)
set -e
TAG=$1; CFG=${2:-c2}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace -d /tmp/prof_$TAG -o res -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 0 --config $CFG --no-cpu-baseline --no-roofline --no-secondary > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
DB=$(find /tmp/prof_$TAG -name "*.db" | head -1)
python3 $GRAFT_REPO_ROOT/scripts/kstats.py $DB 8 > $OUT/${TAG}_kernel_stats.txt
head -60 $OUT/${TAG}_kernel_stats.txt
python3 $GRAFT_REPO_ROOT/scripts/kseq.py $DB 8 > $OUT/${TAG}_kernel_seq.txt
