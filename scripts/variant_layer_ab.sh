#!/bin/bash
# layer times only (no step): variant_layer_ab.sh OUT variant...
R=$GRAFT_REPO_ROOT
V=$R/superresolution_aniso_mri_amd/csrc/build/variants
OUT=$R/gpurun_out/$1.txt
shift
: > $OUT
for rep in 1 2; do
for n in "$@"; do
  export AESR_LIB=$V/libaesr_$n.so
  line="$n:"
  for shape in "36 162 162 32 32" "36 80 80 32 32" "36 40 40 64 64"; do
    T=$(timeout -k 10 100 python3 $R/scripts/time_one.py fwd $shape 2>/dev/null | tail -n 1)
    line="$line  [$shape] $T us"
  done
  echo "$line" | tee -a $OUT
done
done
