#!/bin/bash
# Cache-policy bits on the ring kernel's PATCH DMAs (round-3 verdict, next 5): does a streaming hint (nt) on the activations keep the filter
# chunks in the XCD's L2 across item rounds?  Variant libraries build/variants/libaesr_aux{2,16,18}.so = conv_wino_ring.hip with
# -DRG_PATCH_AUX=2 (nt) / 16 (sc1) / 18 (both); per variant: FETCH_SIZE (x 2, KiB) per launch and the time of 10 back-to-back launches.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_ring_nt.txt
L=$R/superresolution_aniso_mri_amd/libaesr_hip.so
cp $L /tmp/libaesr_keep.so
: > $OUT
for v in 0 2 16 18; do
  if [ $v -gt 0 ]; then cp $R/superresolution_aniso_mri_amd/csrc/build/variants/libaesr_aux$v.so $L; else cp /tmp/libaesr_keep.so $L; fi
  for shape in "24 40 40 256 256" "24 20 20 512 512" "24 80 80 128 128" "36 81 81 64 64"; do
    rm -rf /tmp/rt
    AESR_WINO_RING=2 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/rt -o res -- python3 $R/scripts/bench_one_wino.py fwd $shape 3 > /dev/null 2>&1
    F=$(find /tmp/rt -name "*counter_collection.csv" | head -1)
    T=$(AESR_WINO_RING=2 timeout -k 10 100 python3 $R/scripts/r04_time_one.py fwd $shape 2>/dev/null | tail -n 1)
    python3 - "$F" $v "$T" $shape >> $OUT <<'PY'
import csv, sys
v, T = sys.argv[2], sys.argv[3]
N, H, W, Cin, Cout = [int(x) for x in sys.argv[4:9]]
vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if r["Counter_Name"] == "FETCH_SIZE" and "conv_wino_ring" in r["Kernel_Name"]]
rd = 2 * sum(vals) / max(1, len(vals)) * 1024 / 1e6
inp, flt = N * H * W * Cin * 4 / 1e6, 16 * Cin * Cout * 4 / 1e6
print("aux=%-2s N=%2d %3dx%-3d %3d->%-3d  fabric reads %7.1f MB per launch = %.2f x (input + filter)   %s us" % (v, N, H, W, Cin, Cout, rd, rd / (inp + flt), T))
PY
  done
done
cp /tmp/libaesr_keep.so $L
cat $OUT
