#!/bin/bash
# A/B of conv_wino_res_f32 variants (scripts/variants.py) through AESR_LIB: layer times (HIP events, best of 5 x 10 launches) and the C2 step
R=$GRAFT_REPO_ROOT
V=$R/superresolution_aniso_mri_amd/csrc/build/variants
OUT=$R/gpurun_out/${1:-r05_res_ab}.txt
shift
: > $OUT
for n in "$@"; do
  export AESR_LIB=$V/libaesr_$n.so
  line="$n:"
  for shape in "36 162 162 32 32" "36 80 80 32 32" "36 40 40 64 64" "24 80 80 64 32"; do
    T=$(timeout -k 10 100 python3 $R/scripts/time_one.py fwd $shape 2>/dev/null | tail -n 1)
    line="$line  [$shape] $T us"
  done
  S=$(timeout -k 10 200 python3 $R/bench.py --no-secondary --no-cpu-baseline --no-roofline --steps 40 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  S3=$(timeout -k 10 200 python3 $R/bench.py --config c3 --no-secondary --no-cpu-baseline --no-roofline --steps 20 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$line  | C2 step $S ms  C3 step $S3 ms" | tee -a $OUT
done
