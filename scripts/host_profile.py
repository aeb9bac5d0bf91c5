#!/usr/bin/env python
"""cProfile of host-launched training steps (where does the Python side of an eager step go?): host_profile.py [triplets]
(AESR_FORCE_DP=1: with the data-parallel hooks on an RCCL group of one)."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from superresolution_aniso_mri_amd.data_synth import synthetic_batch  # noqa: E402
from superresolution_aniso_mri_amd.kwatsch.get_trainer import get_trainer_dynamic  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
torch.manual_seed(0)
tr = get_trainer_dynamic(bench.build_args("c2", "cuda:0"))
if os.environ.get("AESR_FORCE_DP") == "1":          # data-parallel call pattern on a process group of one (RCCL)
    from superresolution_aniso_mri_amd.parallel import DataParallelContext
    dp = DataParallelContext(device="cuda:0")
    dp.attach(tr)
    dp.set_batch(B)
pool = [{k: (v.cuda() if torch.is_tensor(v) else v) for k, v in synthetic_batch(B, 160, 160, seed=i).items()} for i in range(4)]
for i in range(5):
    tr.train(pool[i % 4], keep_predictions=False)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(20):
    tr.train(pool[i % 4], keep_predictions=False)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
