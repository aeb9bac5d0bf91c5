#!/bin/bash
# small-shard A/B of an environment knob: the replayed C2 (and C3) step at T triplets, alternating arms, 2 rounds
# usage: shard_env_ab.sh OUTNAME "T T ..." "VAR=a" "VAR=b" ...      (an arm "-" = no variable)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1.txt
TS=$2
shift; shift
: > $OUT
for round in 1 2; do
  for arm in "$@"; do
    E="$arm"; [ "$arm" = "-" ] && E="AESR_NOOP=1"
    line="round $round  $arm "
    for T in $TS; do
      S=$(env $E timeout -k 10 200 python3 $R/bench.py --triplets $T --no-secondary --no-cpu-baseline --no-roofline --steps 60 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
      S3=$(env $E timeout -k 10 200 python3 $R/bench.py --config c3 --triplets $T --no-secondary --no-cpu-baseline --no-roofline --steps 40 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
      line="$line | T=$T C2 $S C3 $S3"
    done
    echo "$line" | tee -a $OUT
  done
done
