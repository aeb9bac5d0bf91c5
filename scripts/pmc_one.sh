#!/bin/bash
# SQ counters of one Winograd layer: pmc_one.sh <tag> <kind> N H W Cin Cout
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC"; do
  rm -rf /tmp/pmc_$TAG
  rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_$TAG -o res -- python3 $GRAFT_REPO_ROOT/scripts/bench_one_wino.py "$@" 3 > /dev/null 2>&1
  F=$(find /tmp/pmc_$TAG -name "*counter_collection.csv" | head -1)
  python3 - "$F" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:40]
    if ("wino" not in k and "wgrad" not in k) or "pack" in k or "reduce" in k: continue
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
done
