#!/bin/bash
# Profile set of one configuration for a round: round_profiles.sh <tag rNN> <cfg c2|c3|c4|c5> [pmc|nopmc] [triplets]
#   kernel statistics + kernel sequence of the REPLAYED (captured graph) step (rocprofv3 --kernel-trace of the bench command) and, with
#   "pmc", the HBM-side traffic per kernel from two --pmc passes (FETCH_SIZE, WRITE_SIZE; host-launched steps).  With a triplet count:
#   the step a RANK of the 8-rank run executes (c4: 2, c5: 1), files <tag>_<cfg>_<T>triplets_*.
set -e
TAG=$1; CFG=$2; PMC=$3; T=$4
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
S=$GRAFT_REPO_ROOT/scripts
NAME=${TAG}_${CFG}; EXTRA=""
if [ -n "$T" ]; then NAME=${TAG}_${CFG}_${T}triplets; EXTRA="--triplets $T"; fi
rm -rf /tmp/rp_$NAME
rocprofv3 --kernel-trace -d /tmp/rp_$NAME/kt -o res -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 4 --config $CFG $EXTRA --no-cpu-baseline --no-roofline --no-secondary > $OUT/${NAME}_prof_bench.json 2> /dev/null
DB=$(find /tmp/rp_$NAME/kt -name "*.db" | head -1)
python3 $S/kstats_last.py $DB 4 "" $OUT/${NAME}_kernel_stats.json > $OUT/${NAME}_kernel_stats.txt
python3 $S/kseq_last.py $DB > $OUT/${NAME}_kernel_sequence.txt
head -14 $OUT/${NAME}_kernel_stats.txt | cut -c1-170
if [ "$PMC" = "pmc" ]; then
  CMD="python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 0 --config $CFG $EXTRA --no-cpu-baseline --no-roofline --no-secondary --no-graph"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/rp_$NAME/f -o res -- $CMD > /dev/null 2>&1
  echo "fetch pass done"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/rp_$NAME/w -o res -- $CMD > /dev/null 2>&1
  echo "write pass done"
  python3 $S/pmc_traffic.py $(find /tmp/rp_$NAME/f -name "*counter_collection.csv" | head -1) $(find /tmp/rp_$NAME/w -name "*counter_collection.csv" | head -1) 5 $OUT/${NAME}_hbm_traffic.json | head -12
fi
