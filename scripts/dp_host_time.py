"""Host time per replayed data-parallel step (communicator of one): is the rank step GPU-bound or launch-bound?
    AESR_FORCE_DP=1 [AESR_DP_GRAPH=whole] python scripts/dp_host_time.py c2 2"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch.distributed as dist
from superresolution_aniso_mri_amd.parallel import DataParallelContext

cfg, t = sys.argv[1], int(sys.argv[2])
dev = "cuda:0"
torch.cuda.set_device(0)
dp = None
if os.environ.get("AESR_FORCE_DP") == "1":
    dist.init_process_group("gloo", store=dist.HashStore(), rank=0, world_size=1)
    dp = DataParallelContext(device=dev)
trainer, pool = bench.make_trainer(cfg, dev, t, 160, npool=2, dp=dp)
for i in range(8):
    trainer.train(pool[i % 2], keep_predictions=False)
torch.cuda.synchronize()
for n in (20, 50, 100, 300, 1000):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        trainer.train(pool[i % 2], keep_predictions=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%s %d triplets dp=%s form=%s, %4d steps: host enqueue %.1f us/step, wall %.1f us/step" % (
        cfg, t, dp is not None, dp.graph_mode if dp else "-", n, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6), flush=True)
if dp is not None:
    dp.shutdown()
