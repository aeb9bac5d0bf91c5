#!/bin/bash
# same-box A/B of an environment knob at FULL batch (12 triplets): the C2 and C3 step, alternating arms, 3 rounds each
# usage: env_ab.sh OUTNAME "VAR=a" "VAR=b" ...
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1.txt
shift
: > $OUT
for round in 1 2 3; do
  for arm in "$@"; do
    S=$(env $arm timeout -k 10 200 python3 $R/bench.py --no-secondary --no-cpu-baseline --no-roofline --steps 40 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    S3=$(env $arm timeout -k 10 200 python3 $R/bench.py --config c3 --no-secondary --no-cpu-baseline --no-roofline --steps 20 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "round $round  $arm  C2 $S ms  C3 $S3 ms" | tee -a $OUT
  done
done
