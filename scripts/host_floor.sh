#!/bin/bash
# host-overhead floor: step time at tiny global batches, host launches vs captured graph
for b in 1 2; do
  for mode in "--no-graph" ""; do
    python bench.py --triplets $b --steps 50 --warmup 8 --no-cpu-baseline --no-roofline $mode 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$b mode=[$mode] ms/step=', d['ms_per_step'])"
  done
done
