#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out
python3 scripts/diag_vggfile_grad.py 48 40 > $OUT/r03_diag_vggfile.txt 2>&1
cat $OUT/r03_diag_vggfile.txt | cut -c1-400
