#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_trainer_api.py tests/test_gpu_step.py tests/test_gpu_ae.py tests/test_gpu_inference.py tests/test_gpu_dp.py -q 2>&1 | grep -E "FAILED|passed|failed|Error" | cut -c1-200
echo "== s3 with the first streamed kernel only"
AESR_WINO_RING=0 timeout -k 10 300 python3 -m pytest "tests/test_gpu_step.py::test_three_train_steps" -q 2>&1 | grep -E "FAILED|passed|failed" | cut -c1-200
bash scripts/small_shards.sh r03_nodes4 c2 2>&1 | tail -8
python3 bench.py --no-secondary > $OUT/bench_r03_try2.json 2> /dev/null; python3 -c "
import json; d=json.load(open('$OUT/bench_r03_try2.json')); r=d['roofline']; print(d['ms_per_step'], r['kernel'], r['achieved'], r['frac'], r['avg_launch_us'], [ (o['kernel'], o['achieved'], o['ms_per_step']) for o in r['other_kernels']])"
