#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 10 500 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_step.py -q -x -k "wgrad or three_train or repeatable or replay" 2>&1 | tail -2 | cut -c1-200
for N in 6 36; do AESR_BENCH_N=$N python3 scripts/bench_conv.py ae 2>/dev/null | grep -E "TOTAL wgrad|dec.12|enc.15"; done
cd /tmp && export TMPDIR=/tmp
for T in 2 12; do rm -rf /tmp/we_$T; rocprofv3 --kernel-trace -d /tmp/we_$T -o res -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 4 --config c2 --triplets $T --no-cpu-baseline --no-roofline --no-secondary > /dev/null 2>&1; python3 $GRAFT_REPO_ROOT/scripts/kstats.py $(find /tmp/we_$T -name "*.db" | head -1) 12 | grep -E "conv_wgrad_wino|total kernel"; done
cd $GRAFT_REPO_ROOT; for T in 1 2 12; do python3 bench.py --steps 40 --warmup 10 --config c2 --triplets $T --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('c2 triplets $T: %.3f ms/step' % d['ms_per_step'])"; done
