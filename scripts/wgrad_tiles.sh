#!/bin/bash
# sweep wgrad tile shapes for one layer: wgrad_tiles.sh N H W Cin Cout "TH,TW TH,TW ..."
for t in $6; do
  echo -n "tile $t: "
  AESR_WGRAD_TILE=$t AESR_WGRAD_DBG=1 python scripts/bench_one.py wgrad $1 $2 $3 $4 $5 2 2>&1 | grep stamps | tail -1 | sed 's/.*kcycles://'
done
