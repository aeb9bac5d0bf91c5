#!/bin/bash
# Round-4 profile set of one configuration: r04_profiles.sh <cfg c2|c3|c4|c5> [pmc]
#   kernel statistics + kernel sequence of the REPLAYED (captured graph) step (rocprofv3 --kernel-trace of the bench command), and with
#   "pmc" the HBM-side traffic per kernel from two --pmc passes (FETCH_SIZE, WRITE_SIZE; host-launched steps, as in rounds 2-3)
set -e
CFG=$1; PMC=$2
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
S=$GRAFT_REPO_ROOT/scripts
rm -rf /tmp/p4_$CFG
rocprofv3 --kernel-trace -d /tmp/p4_$CFG/kt -o res -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 4 --config $CFG --no-cpu-baseline --no-roofline --no-secondary > $OUT/r04_${CFG}_prof_bench.json 2> /dev/null
DB=$(find /tmp/p4_$CFG/kt -name "*.db" | head -1)
python3 $S/kstats_last.py $DB 4 > $OUT/r04_${CFG}_kernel_stats.txt
python3 $S/kseq_last.py $DB > $OUT/r04_${CFG}_kernel_sequence.txt
head -14 $OUT/r04_${CFG}_kernel_stats.txt | cut -c1-170
if [ "$PMC" = "pmc" ]; then
  CMD="python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 0 --config $CFG --no-cpu-baseline --no-roofline --no-secondary --no-graph"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p4_$CFG/f -o res -- $CMD > /dev/null 2>&1
  echo "fetch pass done"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p4_$CFG/w -o res -- $CMD > /dev/null 2>&1
  echo "write pass done"
  python3 $S/pmc_traffic.py $(find /tmp/p4_$CFG/f -name "*counter_collection.csv" | head -1) $(find /tmp/p4_$CFG/w -name "*counter_collection.csv" | head -1) 5 $OUT/r04_${CFG}_hbm_traffic.json | head -12
fi
