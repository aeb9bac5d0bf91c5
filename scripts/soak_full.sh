#!/bin/bash
# full-size soak (soak_full.sh [tag rNN]): every BASELINE configuration's replayed step for 1 000 - 2 000 steps, twice -- identical final losses, no watchdog
OUT=$GRAFT_REPO_ROOT/gpurun_out
TAG=${1:-r04}
F=$OUT/${TAG}_soak_full.txt
: > $F
line() { python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1: %d steps, %.3f ms/step, final loss %.6f, ring watchdog %d, BatchNorm barrier watchdog %d' % (d['steps'], d['ms_per_step'], d['final_loss'], d['ring_watchdog_timeouts'], d['bn_barrier_timeouts']))" >> $F; }
for C in "c2 2000" "c3 1500" "c4 1000" "c5 1000"; do
  set -- $C
  for RUN in 1 2; do
    python3 $GRAFT_REPO_ROOT/bench.py --config $1 --steps $2 --warmup 10 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | line "$1 full batch run $RUN"
  done
done
python3 - >> $F <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch
import bench
from superresolution_aniso_mri_amd import generate_hr_volumes as ghv, _hip
from superresolution_aniso_mri_amd.kwatsch.get_trainer import get_trainer_dynamic
torch.cuda.set_device(0)
torch.manual_seed(892372)
tr = get_trainer_dynamic(bench.build_args("c5", "cuda:0"), eval_mode=True)
vol = torch.rand(30, 1, 224, 224, device="cuda:0")
al = np.linspace(0, 1, 5, endpoint=True)[1:-1]
ref = ghv.create_super_volume(tr, vol, al, use_original=True, to_cpu=False)["upsampled_image"].clone()
same = all(torch.equal(ref, ghv.create_super_volume(tr, vol, al, use_original=True, to_cpu=False)["upsampled_image"]) for _ in range(200))
_hip.check_device_watchdogs("soak")
print("slice synthesis, 30 x 224 x 224, 3 mixes per pair: 200 volumes bitwise equal to the first: %s" % same)
PY
cat $F
