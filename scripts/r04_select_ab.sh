#!/bin/bash
# same-box A/B of the remaining selection knobs at full size (C2, and C3 where LPIPS layers are concerned): ms per step
run() { # label env...
  L=$1; shift
  for c in c2 c3; do
    env "$@" timeout -k 10 200 python3 $GRAFT_REPO_ROOT/bench.py --config $c --steps 40 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s %s %.3f' % ('$L', '$c', d['ms_per_step']))"
  done
}
run "default" X=1
run "AESR_WINO_RES_TN=32" AESR_WINO_RES_TN=32
run "AESR_WINO_RES_TN=16" AESR_WINO_RES_TN=16
run "AESR_WGRAD_WINO_TILE=8,16" AESR_WGRAD_WINO_TILE=8,16
run "AESR_WINO_RES=1 (<=32 ch only)" AESR_WINO_RES=1
run "default again" X=1
