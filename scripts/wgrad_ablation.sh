#!/bin/bash
# Timing-only ablations of conv_wgrad_wino_f32's tile loop (results are wrong by construction): what each ingredient costs per visit.
#   build (here or in the build container):  python scripts/variants.py wabl1=conv_wgrad_wino.hip:-DWW_ABL=1 wabl2=...:-DWW_ABL=2 wabl3=...:-DWW_ABL=3 wabl4=...:-DWW_ABL=4
#   WW_ABL: 1 no DMA, 2 no transforms, 3 no LDS reads, 4 MFMAs only.  Variants are loaded through AESR_LIB: the shipped library is not touched.
R=${GRAFT_REPO_ROOT:-.}
V=$R/superresolution_aniso_mri_amd/csrc/build/variants
for n in 0 1 2 3 4; do
  echo "== WW_ABL=$n"
  if [ $n -gt 0 ]; then export AESR_LIB=$V/libaesr_wabl$n.so; else unset AESR_LIB; fi
  AESR_WGRAD_WINO_DBG=1 timeout -k 10 120 python3 $R/scripts/bench_one_wino.py wgrad 36 160 160 32 32 1 2>&1 | grep stamps
done
