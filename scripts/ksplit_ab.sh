#!/bin/bash
# A/B: the ring kernel's channel split on small shards (AESR_RING_KSPLIT=1 forces no split)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_ksplit_ab.txt
: > $OUT
for cfg in "c5 1" "c4 2" "c3 2" "c2 2"; do
  set -- $cfg
  for ks in default 1; do
    if [ $ks = default ]; then unset AESR_RING_KSPLIT; else export AESR_RING_KSPLIT=$ks; fi
    for rep in 1 2; do
      ms=$(python3 bench.py --config $1 --triplets $2 --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
      echo "$1 $2 triplets, AESR_RING_KSPLIT=$ks, run $rep: $ms ms/step" >> $OUT
    done
  done
done
cat $OUT
