#!/bin/bash
# Timing-only ablations of conv_wgrad_wino_f32's tile loop (results are wrong by construction): what each ingredient costs per visit.
# Variant libraries are built by hand with -DWW_ABL=n into superresolution_aniso_mri_amd/csrc/build/variants/ (see DESIGN.md section 5).
L=superresolution_aniso_mri_amd/libaesr_hip.so
cp $L /tmp/libaesr_keep.so
for n in 0 1 2 3 4; do
  if [ $n -gt 0 ]; then cp superresolution_aniso_mri_amd/csrc/build/variants/libaesr_abl$n.so $L; fi
  echo "== WW_ABL=$n"
  AESR_WGRAD_WINO_DBG=1 timeout -k 10 120 python scripts/bench_one_wino.py wgrad 36 160 160 32 32 1 2>&1 | grep stamps
done
cp /tmp/libaesr_keep.so $L
