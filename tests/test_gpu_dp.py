"""-m gpu: the data-parallel training step end to end with TWO ranks sharing the one GPU of the test box (gloo backend,
device tensors staged through the host): triplet sharding with uneven shards (1 + 2), SyncBN (global batch statistics),
w_r-weighted loss and the flat gradient all-reduce must reproduce the single-process step."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _args(mix):
    from superresolution_aniso_mri_amd.networks.net_config import NetworkConfig
    a = dict(model="ae_combined", dataset="OASIS", device="cuda:0", lr=1e-3, weight_decay=0.0, epochs=5, width=32, latent_width=8, depth=8,
             latent=16, ex_loss_weight1=0.05, use_percept_loss=False, get_masks=False, use_loss_annealing=False,
             use_extra_latent_loss=False, epoch_threshold=100, ae_class="VanillaACAI", image_mix_loss_func=mix,
             vgg_weights="synthetic-hash")
    for k, v in NetworkConfig("ae_combined", dataset="OASIS").architecture.items():
        a.setdefault(k, v)
    return a


def _worker(rank, world, port, B, mix, out, graph=False, steps=2):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import warnings
    warnings.simplefilter("ignore")
    from superresolution_aniso_mri_amd.data_synth import shard_batch, synthetic_batch
    from superresolution_aniso_mri_amd.kwatsch.get_trainer import get_trainer_dynamic
    from superresolution_aniso_mri_amd.parallel import DataParallelContext
    torch.cuda.set_device(0)
    dp = DataParallelContext(backend="gloo", device="cuda:0")
    torch.manual_seed(100 + rank)                       # ranks start from different weights; attach() broadcasts rank 0's
    tr = get_trainer_dynamic(_args(mix))
    dp.attach(tr)
    dp.set_batch(B)
    if graph:
        tr.enable_step_graph(eager_steps=1, dp_segments=True)      # step 0 eager, step 1 captured segment by segment, then replays
    for step in range(steps):
        tr.train(shard_batch(synthetic_batch(B, 32, 32, seed=40 + step, brain=True), rank, world), keep_predictions=False)
    if rank == 0:
        torch.save({"sd": {k: v.cpu() for k, v in tr.model.state_dict().items()},
                    "loss": [dp.reduce_scalar(v) for v in tr.losses["loss_ae"].floats()]}, out)
    else:
        [dp.reduce_scalar(v) for v in tr.losses["loss_ae"].floats()]
    dp.barrier()


@pytest.mark.parametrize("mix", ["mse", "perceptual"])
def test_two_ranks_equal_single_process(tmp_path, mix):
    import warnings
    from superresolution_aniso_mri_amd.data_synth import synthetic_batch
    from superresolution_aniso_mri_amd.kwatsch.get_trainer import get_trainer_dynamic
    B = 3
    out = str(tmp_path / "dp.pt")
    mp.spawn(_worker, args=(2, _free_port(), B, mix, out), nprocs=2, join=True)
    res = torch.load(out)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.manual_seed(100)
        ref = get_trainer_dynamic(_args(mix))
    for step in range(2):
        ref.train(synthetic_batch(B, 32, 32, seed=40 + step, brain=True), keep_predictions=False)
    np.testing.assert_allclose(res["loss"], ref.losses["loss_ae"].floats(), rtol=2e-5)
    for k, v in ref.model.state_dict().items():
        a, b = res["sd"][k].double(), v.cpu().double()
        if "num_batches" in k:
            assert int(a) == int(b)
            continue
        diff = (a - b).abs()
        # two Adam steps at lr 1e-3: identical up to summation order (sign noise of ~0 gradients allowed on a few elements)
        assert float(diff.max()) <= 2 * 2 * 1e-3 + 1e-6, k
        assert float((diff > 1e-4 + 1e-3 * b.abs()).double().mean()) <= 0.03, k


def test_segmented_step_graph_equals_host_launched_data_parallel(tmp_path):
    """Data parallel with the step captured as a chain of HIP graphs cut at the (eager) collectives
    (parallel.SegmentedStepGraph) == the host-launched data-parallel step: 5 steps (1 eager, 1 capture, 3 replays), two ranks
    on one GPU, losses and final weights."""
    B = 3
    outs = []
    for graph in (False, True):
        out = str(tmp_path / ("dp_%d.pt" % graph))
        mp.spawn(_worker, args=(2, _free_port(), B, "mse", out, graph, 5), nprocs=2, join=True)
        outs.append(torch.load(out))
    eager, seg = outs
    np.testing.assert_allclose(seg["loss"], eager["loss"], rtol=1e-6)
    for k, v in eager["sd"].items():
        if v.dtype.is_floating_point:
            assert float((seg["sd"][k].double() - v.double()).abs().max()) <= 1e-6 + 1e-5 * float(v.abs().max()), k
        else:
            assert int(seg["sd"][k]) == int(v)
