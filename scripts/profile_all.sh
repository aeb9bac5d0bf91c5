#!/bin/bash
# Round profile set: kernel statistics (rocprofv3 --kernel-trace) and HBM-side traffic (two --pmc passes) of the bench step.
#   profile_all.sh <round tag, e.g. r02> <config c2|c3>      -> gpurun_out/<tag>_<cfg>_kernel_stats.txt, <tag>_<cfg>_hbm_traffic.json
set -e
TAG=$1; CFG=${2:-c2}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
CMD="python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 0 --config $CFG --no-cpu-baseline --no-roofline --no-secondary --no-graph"
rm -rf /tmp/pa_$TAG
rocprofv3 --kernel-trace -d /tmp/pa_$TAG/kt -o res -- $CMD > $OUT/${TAG}_${CFG}_prof_bench.json 2> /dev/null
python3 $GRAFT_REPO_ROOT/scripts/kstats.py $(find /tmp/pa_$TAG/kt -name "*.db" | head -1) 5 > $OUT/${TAG}_${CFG}_kernel_stats.txt
echo "kernel-trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pa_$TAG/f -o res -- $CMD > /dev/null 2>&1
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pa_$TAG/w -o res -- $CMD > /dev/null 2>&1
echo "write pass done"
python3 $GRAFT_REPO_ROOT/scripts/pmc_traffic.py $(find /tmp/pa_$TAG/f -name "*counter_collection.csv" | head -1) $(find /tmp/pa_$TAG/w -name "*counter_collection.csv" | head -1) 5 $OUT/${TAG}_${CFG}_hbm_traffic.json | head -14
head -12 $OUT/${TAG}_${CFG}_kernel_stats.txt
