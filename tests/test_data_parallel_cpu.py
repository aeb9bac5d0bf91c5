"""not-gpu: the N>1 path with world_size 2 on the gloo backend -- triplet sharding with uneven shards, the
w_r-weighted gradient all-reduce, SyncBN partial-sum all-reduce and scalar reductions of
superresolution_aniso_mri_amd/parallel.py.  Compute is done by the CPU oracle (test infrastructure)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from oracle import ae_oracle
    from superresolution_aniso_mri_amd.data_synth import shard_batch, synthetic_batch
    from superresolution_aniso_mri_amd.parallel import DataParallelContext
    dp = DataParallelContext(backend="gloo", device="cpu")
    lo, hi = dp.set_batch(B)
    cfg = dict(width=32, latent_width=8, depth=8, latent=16, colors=1, use_batchnorm=True, use_sigmoid=True)
    torch.manual_seed(100 + rank)                      # deliberately different initial weights per rank
    ae = ae_oracle.OracleAE(cfg)

    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.ps = torch.nn.ParameterList([torch.nn.Parameter(p.detach().clone()) for p in ae.parameters()])
            for k, b in ae.buffers.items():
                self.register_buffer(k.replace(".", "_"), b.clone())

    holder = Holder()
    dp.broadcast_parameters(holder)                     # rank 0's weights everywhere
    for p, q in zip(ae.parameters(), holder.ps):
        p.data.copy_(q.data)
    full = synthetic_batch(B, 32, 32, seed=5)
    mine = shard_batch(full, rank, world)
    assert mine["slice_between"].shape[0] == hi - lo
    # eval-mode BatchNorm: per-sample independent -> sharded == single process exactly (up to fp order)
    out_local = ae.forward(mine["image"], train=False)
    loss = F.mse_loss(out_local, mine["image"])
    (loss * dp.weight).backward()
    opt = torch.optim.SGD(ae.parameters(), lr=0.0)
    dp.allreduce_gradients(opt)
    gl = dp.reduce_scalar(float(loss), weighted=True)
    # SyncBN hook on fake partial sums
    sums = torch.full((1, 2, 4), float(rank + 1), dtype=torch.float64)
    dp.sync_bn(sums)
    assert float(sums[0, 0, 0]) == sum(range(1, world + 1))
    assert abs((hi - lo) / dp.weight - B) < 1e-9          # local count * count_scale == global count
    assert dp.max_over_ranks(rank) == world - 1
    if rank == 0:
        torch.save({"grads": [p.grad.clone() for p in ae.parameters()], "loss": gl,
                    "params": [p.detach().clone() for p in ae.parameters()]}, out)
    dp.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [3, 4])
def test_two_rank_gloo_matches_single_process(tmp_path, B):
    from oracle import ae_oracle
    from superresolution_aniso_mri_amd.data_synth import synthetic_batch
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), B, out), nprocs=2, join=True)
    res = torch.load(out)
    cfg = dict(width=32, latent_width=8, depth=8, latent=16, colors=1, use_batchnorm=True, use_sigmoid=True)
    torch.manual_seed(100)
    ae = ae_oracle.OracleAE(cfg)                        # == rank 0's weights
    for p, q in zip(ae.parameters(), res["params"]):
        assert torch.equal(p.detach(), q)
    full = synthetic_batch(B, 32, 32, seed=5)
    loss = F.mse_loss(ae.forward(full["image"], train=False), full["image"])
    loss.backward()
    assert abs(res["loss"] - float(loss)) < 1e-6 * float(loss)      # B=3: shards 1 + 2 triplets, weights 1/3 and 2/3
    for p, g in zip(ae.parameters(), res["grads"]):
        np.testing.assert_allclose(g.numpy(), p.grad.numpy(), rtol=2e-4, atol=1e-9)


@pytest.mark.parametrize("B", [12, 8])
def test_eight_rank_gloo_matches_single_process(tmp_path, B):
    """The shard arithmetic of the 8-GPU runs the driver launches, rehearsed on the CPU: 12 triplets over 8 ranks (BASELINE configs[1] / [2]:
    uneven shards 1,2,1,2,1,2,1,2 with weights B_r / B) and 8 over 8 (configs[4]: one triplet per rank) -- weighted flat gradient all-reduce,
    global loss, SyncBN hook, parameter broadcast, max-over-ranks equal the single-process values."""
    from oracle import ae_oracle
    from superresolution_aniso_mri_amd.data_synth import synthetic_batch
    sizes = [(B * (r + 1)) // 8 - (B * r) // 8 for r in range(8)]           # DataParallelContext.shard_range / data_synth.shard_batch
    assert sum(sizes) == B and min(sizes) >= 1 and max(sizes) - min(sizes) <= 1, sizes
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(8, _free_port(), B, out), nprocs=8, join=True)
    res = torch.load(out)
    cfg = dict(width=32, latent_width=8, depth=8, latent=16, colors=1, use_batchnorm=True, use_sigmoid=True)
    torch.manual_seed(100)
    ae = ae_oracle.OracleAE(cfg)
    full = synthetic_batch(B, 32, 32, seed=5)
    loss = F.mse_loss(ae.forward(full["image"], train=False), full["image"])
    loss.backward()
    assert abs(res["loss"] - float(loss)) < 1e-6 * float(loss)
    for p, g in zip(ae.parameters(), res["grads"]):
        np.testing.assert_allclose(g.numpy(), p.grad.numpy(), rtol=2e-4, atol=1e-9)


def _log_worker(rank, world, port, B, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from collections import defaultdict
    from superresolution_aniso_mri_amd.kwatsch.base_trainer import BaseTrainer, LossLog
    from superresolution_aniso_mri_amd.parallel import DataParallelContext
    dp = DataParallelContext(backend="gloo", device="cpu")
    lo, hi = dp.set_batch(B)
    # per-triplet "losses" 1..B; every rank logs the mean over ITS shard, twice (two iterations of an epoch)
    per = np.arange(1, B + 1, dtype=np.float64)
    t = BaseTrainer()
    t.args, t.dp, t._iters = {}, dp, 7
    t.losses, t.losses_test = defaultdict(LossLog), defaultdict(LossLog)
    t.mean_losses, t.mean_losses_test, t.loss_iters = defaultdict(list), defaultdict(list), []
    for scale in (1.0, 3.0):
        t.losses["loss_ae"].append(float(per[lo:hi].mean() * scale))
        t.losses_test["loss_ae_dist"].append(float((per[lo:hi] ** 2).mean() * scale))
    t._log_count.update(train=hi - lo, test=hi - lo)
    t.show_loss_on_tensorboard()
    t.show_loss_on_tensorboard(eval_type="test")
    if rank == 0:
        torch.save({"train": t.mean_losses["loss_ae"][-1], "test": t.mean_losses_test["loss_ae_dist"][-1]}, out)
    dp.barrier()
    dist.destroy_process_group()


def test_logged_means_are_global_under_data_parallel(tmp_path):
    """Model selection and the loss files see the mean over the GLOBAL batch, not rank 0's shard (uneven shards 1 + 2)."""
    out = str(tmp_path / "log.pt")
    B = 3
    mp.spawn(_log_worker, args=(2, _free_port(), B, out), nprocs=2, join=True)
    res = torch.load(out)
    per = np.arange(1, B + 1, dtype=np.float64)
    assert abs(res["train"] - per.mean() * 2.0) < 1e-12
    assert abs(res["test"] - (per ** 2).mean() * 2.0) < 1e-12
