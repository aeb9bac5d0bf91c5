#!/bin/bash
# Where the ring kernel's fabric reads come from (round-3 verdict, next 5: "check with the PMC pass which of the two dominates before touching
# code"): FETCH_SIZE (x 2: gfx950 correction, KiB) of conv_wino_ring_f32 on VGG-shaped layers, with the output channels cut down to ONE cout
# tile (32) -- then every input patch is fetched by exactly one workgroup pass and the filter is 1/8 -- against the full layer.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_ring_traffic.txt
: > $OUT
run() {   # N H W Cin Cout
  rm -rf /tmp/rt
  AESR_WINO_RING=2 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/rt -o res -- python3 $GRAFT_REPO_ROOT/scripts/bench_one_wino.py fwd $1 $2 $3 $4 $5 3 > /dev/null 2>&1
  F=$(find /tmp/rt -name "*counter_collection.csv" | head -1)
  python3 - "$F" $1 $2 $3 $4 $5 >> $OUT <<'PY'
import csv, sys
N, H, W, Cin, Cout = [int(v) for v in sys.argv[2:7]]
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if r["Counter_Name"] == "FETCH_SIZE" and "conv_wino_ring" in r["Kernel_Name"]]
rd = 2 * sum(v) / max(1, len(v)) * 1024 / 1e6
inp, flt = N * H * W * Cin * 4 / 1e6, 16 * Cin * Cout * 4 / 1e6
print("N=%2d %3dx%-3d %3d->%-3d  fabric reads %7.1f MB per launch | input %6.1f MB (x 1.56 halo = %6.1f), transformed filter %5.1f MB (x 8 XCDs = %5.1f) | reads / (input + filter) = %.2f"
      % (N, H, W, Cin, Cout, rd, inp, 1.5625 * inp, flt, 8 * flt, rd / (inp + flt)))
PY
}
for C in 32 64 128 256; do run 24 40 40 256 $C; done
for C in 32 128; do run 24 80 80 128 $C; done
for C in 32 512; do run 24 20 20 512 $C; done
run 24 160 160 64 64
cat $OUT
