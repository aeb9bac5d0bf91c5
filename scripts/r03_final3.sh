#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python3 -m pytest tests -m gpu -q > $OUT/r03_gpu_tests.txt 2>&1; tail -3 $OUT/r03_gpu_tests.txt | cut -c1-250
bash scripts/profile_all.sh r03 c2 > $OUT/r03_prof_c2.log 2>&1; echo "c2 profile rc $?"
bash scripts/profile_all.sh r03 c3 > $OUT/r03_prof_c3.log 2>&1; echo "c3 profile rc $?"
cd $GRAFT_REPO_ROOT
python3 bench.py > $OUT/r03_bench_try4.json 2> $OUT/r03_bench_try4.err; echo "bench rc $?"; tail -c 400 $OUT/r03_bench_try4.err
