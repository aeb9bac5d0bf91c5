#!/bin/bash
# Kernel statistics of bench.py's inference leg (generate_hr_volumes.create_super_volume, 30 x 224 x 224, 3 interpolations, dHCP model)
set -e
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
rm -rf /tmp/p4_inf
rocprofv3 --kernel-trace -d /tmp/p4_inf -o res -- python3 $GRAFT_REPO_ROOT/scripts/infer_leg.py > $OUT/${TAG:-r05}_inference_leg.json 2> /dev/null
DB=$(find /tmp/p4_inf -name "*.db" | head -1)
python3 $GRAFT_REPO_ROOT/scripts/kstats.py $DB 7 > $OUT/${TAG:-r05}_inference_kernel_stats.txt
head -16 $OUT/${TAG:-r05}_inference_kernel_stats.txt | cut -c1-170
