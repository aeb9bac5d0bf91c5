#!/bin/bash
# Timing-only ablations of conv_wino_res_f32 (results are wrong by construction), as scripts/wgrad_ablation.sh did for the weight gradient:
# variant libraries build/variants/libaesr_resabl{0..5}.so = a scratch copy of conv_wino_res.hip with -DWR_ABL=n:
#   0 as shipped | 1 no patch DMAs after the first | 2 no input transform | 3 no LDS reads of patch and filter in the chunk loop | 4 = 1 + 2 + 3 + 5 (MFMAs only) | 5 no epilogue
R=$GRAFT_REPO_ROOT
L=$R/superresolution_aniso_mri_amd/libaesr_hip.so
cp $L /tmp/libaesr_keep.so
for n in 0 1 2 3 5 4; do
  cp $R/superresolution_aniso_mri_amd/csrc/build/variants/libaesr_resabl$n.so $L
  for shape in "36 162 162 32 32" "36 80 80 32 32" "36 40 40 64 64"; do
    T=$(timeout -k 10 100 python3 $R/scripts/r04_time_one.py fwd $shape 2>/dev/null | tail -n 1)
    echo "WR_ABL=$n  fwd $shape  $T us"
  done
done
cp /tmp/libaesr_keep.so $L
