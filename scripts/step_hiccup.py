"""Per-step host time of the replayed step: where are the outliers?   python scripts/step_hiccup.py c2 2"""
import gc, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
cfg, t = sys.argv[1], int(sys.argv[2])
torch.cuda.set_device(0)
trainer, pool = bench.make_trainer(cfg, "cuda:0", t, 160, npool=2)
for i in range(8):
    trainer.train(pool[i % 2], keep_predictions=False)
torch.cuda.synchronize()
if len(sys.argv) > 3 and sys.argv[3] == "nogc":
    gc.disable()
ts = []
for i in range(1500):
    t0 = time.perf_counter()
    trainer.train(pool[i % 2], keep_predictions=False)
    if i % 8 == 7:
        torch.cuda.synchronize()          # keep the queue shallow: a host-side stall shows as itself, not as back-pressure
    ts.append(time.perf_counter() - t0)
out = [(i, round(v * 1e3, 2)) for i, v in enumerate(ts) if v > 5e-3]
print("gc", gc.isenabled(), "outliers > 5 ms (step, ms):", out[:20], "| median %.1f us" % (sorted(ts)[len(ts) // 2] * 1e6), "| reserved MB", torch.cuda.memory_reserved() >> 20)
