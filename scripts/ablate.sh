#!/bin/bash
# ablation of the igemm kernel on one layer: AESR_IGEMM_DBG bit0 no patch loads, bit1 no weight loads, bit2 no stores
for d in 0 1 2 4 3 7; do
  echo "== dbg=$d"; AESR_IGEMM_DBG=$d python scripts/bench_conv.py ae 2>/dev/null | grep -E "enc.1/3|enc.15|dec.12" | cut -c1-75
done
