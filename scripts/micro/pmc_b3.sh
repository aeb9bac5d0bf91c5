#!/bin/bash
# PMC passes over the b3v2 driver (timing-only mode): where do the waves of the bf16 x 3 kernel and of conv_wino_res_f32 spend their cycles
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_b3
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export B3_TIME_ONLY=1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace -d $OUT/p1 --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/micro/wino_b3v2.py > $OUT/p1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA --kernel-trace -d $OUT/p2 --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/micro/wino_b3v2.py > $OUT/p2.log 2>&1 || exit 1
python3 $GRAFT_REPO_ROOT/scripts/micro/pmc_b3_summary.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
find $OUT -name "*.csv" -size +2M -delete
