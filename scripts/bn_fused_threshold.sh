#!/bin/bash
# where does the one-launch BatchNorm stop paying?  C2 step at 1 / 2 / 3 / 6 / 12 triplets under thresholds on the layer's bytes (AESR_BN_FUSED_MAX_MB)
R=$GRAFT_REPO_ROOT
# (the 9-image rule of engine.bn_fused_pays would keep every call at 4 and more triplets on the three-launch path whatever the byte threshold:
#  lifted here so that the arms differ; same-arm repeats of this sweep differ by up to ~10 us on 1-2.3 ms steps)
export AESR_BN_FUSED_MAX_IMAGES=1000
OUT=$R/gpurun_out/r05_bn_fused_threshold.txt
echo "C2 step, ms, replayed graph (bench.py --triplets T --no-secondary --steps 60), AESR_BN_FUSED_MAX_MB = threshold on N*H*W*C*4 (0 = three launches everywhere, 1000 = wherever the layer fits LDS)" > $OUT
echo "layer bytes at T triplets: enc.5 10.1 T MB, enc.11 5.0 T MB, dec.4 1.2 T MB, dec.10 2.5 T MB" >> $OUT
for T in 1 2 3 6 12; do
  line="T=$T:"
  for mb in 0 4 8 12 16 24 1000 0 1000; do
    S=$(AESR_BN_FUSED_MAX_MB=$mb timeout -k 10 200 python3 $R/bench.py --triplets $T --no-secondary --no-cpu-baseline --no-roofline --steps 60 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    line="$line  [$mb MB] $S"
  done
  echo "$line" | tee -a $OUT
done
