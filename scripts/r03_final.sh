#!/bin/bash
# round-3 profile set at the final kernel sources, one box
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
bash scripts/profile_all.sh r03 c2 > $OUT/r03_prof_c2.log 2>&1; echo "c2 profile rc $?"
bash scripts/profile_all.sh r03 c3 > $OUT/r03_prof_c3.log 2>&1; echo "c3 profile rc $?"
cd $GRAFT_REPO_ROOT
bash scripts/small_shards.sh r03 c2 > /dev/null 2>&1; echo "small c2 rc $?"
bash scripts/small_shards.sh r03b c3 > /dev/null 2>&1; echo "small c3 rc $?"
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 scripts/bench_wino.py vgg > $OUT/r03_wino_layers_vgg.txt 2>&1; echo "wino vgg rc $?"
timeout -k 10 300 python3 scripts/bench_wino.py ae > $OUT/r03_wino_layers_ae.txt 2>&1; echo "wino ae rc $?"
timeout -k 10 300 python3 tests/diag_relu_flips.py > $OUT/r03_gradient_flip_analysis.txt 2>&1; echo "flips rc $?"
timeout -k 10 900 python3 tests/parity_report.py c1 c2 c3 c4 c5 > $OUT/r03_parity_report.txt 2>&1; echo "parity rc $?"
tail -3 $OUT/r03_wino_layers_vgg.txt | cut -c1-220
