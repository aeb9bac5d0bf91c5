#!/usr/bin/env python
"""Rehearsal of the data-parallel call pattern on ONE GPU with the real RCCL backend (a process group of one, AESR_FORCE_DP=1):
SyncBN + gradient all-reduces through parallel.DataParallelContext: host-launched, as a segmented step graph
(parallel.SegmentedStepGraph, collectives eager) and as one graph with the collectives captured.  Prints ms/step of each and
checks that the losses agree.
    AESR_FORCE_DP=1 python scripts/dp_graph_smoke.py [triplets]"""
import os
import sys
import time

os.environ.setdefault("AESR_FORCE_DP", "1")
os.environ.setdefault("MASTER_PORT", "29533")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from superresolution_aniso_mri_amd.data_synth import synthetic_batch  # noqa: E402
from superresolution_aniso_mri_amd.kwatsch.get_trainer import get_trainer_dynamic  # noqa: E402
from superresolution_aniso_mri_amd.parallel import DataParallelContext  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
torch.cuda.set_device(0)
dp = DataParallelContext(device="cuda:0")
assert dp.active
res = {}
MODES = [m for m in ("host-launched", "segmented graph", "whole graph") if m.split()[0] in os.environ.get("MODES", "host-launched,segmented,whole")]
for mode in MODES:
    torch.manual_seed(892372)
    tr = get_trainer_dynamic(bench.build_args("c2", "cuda:0"))
    dp.attach(tr)
    dp.set_batch(B)
    if mode != "host-launched":
        tr.enable_step_graph(eager_steps=2, dp_mode="segments" if mode.startswith("segmented") else "whole")
    pool = [{k: (v.cuda() if torch.is_tensor(v) else v) for k, v in synthetic_batch(B, 160, 160, seed=892372 + i).items()} for i in range(4)]
    for i in range(6):
        tr.train(pool[i % 4], keep_predictions=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 30
    for i in range(n):
        tr.train(pool[(6 + i) % 4], keep_predictions=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    res[mode] = (dt, tr.losses["loss_ae"].floats()[-1])
    print("%-16s %6.3f ms/step  (%d triplets, RCCL group of one: %d collectives per step)  final loss %.6f"
          % (mode, dt * 1e3, B, 17, res[mode][1]))
for m in MODES[1:]:
    assert abs(res[MODES[0]][1] - res[m][1]) < 1e-6 * abs(res[MODES[0]][1]) + 1e-9, m
print("ok")
torch.distributed.destroy_process_group()
