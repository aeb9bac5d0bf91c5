#!/bin/bash
timeout -k 10 600 python3 -m pytest tests/test_gpu_lpips.py tests/test_gpu_ae.py::test_stem_folded_pass_equals_unfolded_pass tests/test_gpu_trainer_api.py::test_vgg_weights_file_is_loaded "tests/test_gpu_baseline_parity.py::test_first_step_against_the_reference_trainer_probe" -q 2>&1 | grep -E "FAILED|passed|failed|Error" | cut -c1-200
for F in 0 1 0 1; do
AESR_LPIPS_FOLD=$F python3 bench.py --steps 20 --warmup 6 --config c3 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('lpips conv1_1 fold $F, c3: %.3f ms/step loss %.6f' % (d['ms_per_step'], d['final_loss']))"
done
