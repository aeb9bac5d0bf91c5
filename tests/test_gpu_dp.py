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


def _worker(rank, world, port, B, mix, out, graph=False, steps=2, syncbn="rccl"):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), AESR_SYNCBN=syncbn)
    if syncbn == "p2p":
        # several ranks REHEARSED on one device: their one-launch BatchNorm kernels wait for each other's records, so all of them must be
        # resident together -- 64 workgroups each instead of one per CU
        os.environ["AESR_BN_FUSED_NB"] = "64"
        os.environ["AESR_P2P_SPINS"] = str(1 << 18)          # a peer that never arrives costs ~1 s here, not 30
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
    torch.cuda.synchronize()
    from superresolution_aniso_mri_amd import _hip
    _hip.check_device_watchdogs("data-parallel worker %d" % rank)
    if rank == 0:
        torch.save({"sd": {k: v.cpu() for k, v in tr.model.state_dict().items()}, "ncoll": dp.n_collectives,
                    "loss": [dp.reduce_scalar(v) for v in tr.losses["loss_ae"].floats()]}, out)
    else:
        [dp.reduce_scalar(v) for v in tr.losses["loss_ae"].floats()]
    dp.barrier()
    dp.shutdown()


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


@pytest.mark.parametrize("graph", [False, True])
def test_two_ranks_peer_exchange_on_one_device(tmp_path, graph):
    """AESR_SYNCBN=p2p (parallel.PeerExchange, csrc/p2p.hip): two ranks rehearsed on ONE device map each other's exchange region through IPC
    handles; the SyncBN sums of the BatchNorm calls that fit the one-launch kernel travel as direct writes into the peer's region inside
    that kernel (uneven shards: 1 + 2 triplets), the others and the gradients through the host-staged gloo data plane.  Must reproduce
    the single-process step, host-launched and as graph segments; fewer collectives than the all-reduce form."""
    import warnings
    from superresolution_aniso_mri_amd.data_synth import synthetic_batch
    from superresolution_aniso_mri_amd.kwatsch.get_trainer import get_trainer_dynamic
    B, steps = 3, 4
    out = str(tmp_path / "p2p.pt")
    mp.spawn(_worker, args=(2, _free_port(), B, "mse", out, graph, steps, "p2p"), nprocs=2, join=True)
    res = torch.load(out)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.manual_seed(100)
        ref = get_trainer_dynamic(_args("mse"))
    for step in range(steps):
        ref.train(synthetic_batch(B, 32, 32, seed=40 + step, brain=True), keep_predictions=False)
    np.testing.assert_allclose(res["loss"], ref.losses["loss_ae"].floats(), rtol=2e-5)
    for k, v in ref.model.state_dict().items():
        a, b = res["sd"][k].double(), v.cpu().double()
        if "num_batches" in k:
            assert int(a) == int(b)
            continue
        diff = (a - b).abs()
        assert float(diff.max()) <= steps * 2 * 1e-3 + 1e-6, k
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


def _rccl_worker(rank, port, out, graph_form, syncbn="rccl"):
    """A data-parallel group of ONE on the library-owned RCCL communicator (RCCL refuses two ranks on one device, so a one-GPU box
    exercises the call pattern with nranks = 1): raw collectives, then the trainer step host-launched and captured -- as graph
    segments between eager RCCL enqueues (the default form) or as ONE graph with the RCCL calls as nodes (AESR_DP_GRAPH=whole)."""
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), AESR_FORCE_DP="1",
                      AESR_DP_GRAPH=graph_form, AESR_SYNCBN=syncbn)
    os.environ.pop("AESR_DIST_BACKEND", None)
    import ctypes
    import warnings
    warnings.simplefilter("ignore")
    from superresolution_aniso_mri_amd import _hip
    from superresolution_aniso_mri_amd.data_synth import synthetic_batch
    from superresolution_aniso_mri_amd.kwatsch.get_trainer import get_trainer_dynamic
    from superresolution_aniso_mri_amd.parallel import DataParallelContext
    torch.cuda.set_device(0)
    dp = DataParallelContext(device="cuda:0")
    assert dp.active and dp.data_backend == "rccl" and dp.graph_mode == graph_form
    ver = ctypes.c_int(0)
    _hip.check(_hip.lib.aesr_comm_rccl_version(ctypes.byref(ver)), "aesr_comm_rccl_version")
    res = {"rccl_version": ver.value}
    # raw collectives through the C ABI
    a32 = torch.arange(1000, device="cuda", dtype=torch.float32)
    a64 = torch.arange(77, device="cuda", dtype=torch.float64) * 0.5
    dp._all_reduce(a32)
    dp._all_reduce(a64, "max")
    bufs = [torch.full((n,), float(n), device="cuda") for n in (3, 0, 130)]
    arr = (ctypes.c_void_p * 3)(*[b.data_ptr() for b in bufs])
    cnt = (ctypes.c_size_t * 3)(*[b.numel() for b in bufs])
    _hip.check(_hip.lib.aesr_comm_allreduce_many(dp.ensure_comm(), arr, cnt, 3, _hip.COMM_F32, _hip.COMM_SUM, _hip.stream()), "allreduce_many")
    torch.cuda.synchronize()
    res["raw_ok"] = bool(torch.equal(a32.cpu(), torch.arange(1000, dtype=torch.float32)) and
                         torch.equal(a64.cpu(), torch.arange(77, dtype=torch.float64) * 0.5) and float(bufs[2][0]) == 130.0)
    # argument errors are reported, not crashed on
    rc = _hip.lib.aesr_comm_allreduce(dp.ensure_comm(), _hip.ptr(a32), 4, 7, 0, _hip.stream())
    res["bad_dtype"] = (rc, _hip.last_error())
    torch.manual_seed(100)
    tr = get_trainer_dynamic(_args("mse"))
    dp.attach(tr)
    dp.set_batch(3)
    tr.enable_step_graph(eager_steps=2)                  # dp_mode picked from the data plane / AESR_DP_GRAPH
    n0 = dp.n_collectives
    for step in range(5):                                # 2 host-launched, 1 capture, 2 replays
        tr.train(synthetic_batch(3, 32, 32, seed=40 + step, brain=True), keep_predictions=False)
        if step == 0:
            res["collectives_per_step"] = dp.n_collectives - n0
    torch.cuda.synchronize()
    _hip.check_device_watchdogs("one-rank data-parallel worker")
    res.update(graphs=len(tr._graphs), graph_dp=tr._graph_dp, loss=tr.losses["loss_ae"].floats(),
               sd={k: v.cpu() for k, v in tr.model.state_dict().items()})
    torch.save(res, out)
    dp.shutdown()


@pytest.mark.parametrize("syncbn", ["rccl", "p2p"])
@pytest.mark.parametrize("graph_form", ["segments", "whole"])
def test_rccl_communicator_and_step_graph_forms(tmp_path, graph_form, syncbn):
    import warnings
    from superresolution_aniso_mri_amd.data_synth import synthetic_batch
    from superresolution_aniso_mri_amd.kwatsch.get_trainer import get_trainer_dynamic
    out = str(tmp_path / "rccl.pt")
    mp.spawn(_rccl_worker, args=(_free_port(), out, graph_form, syncbn), nprocs=1, join=True)
    res = torch.load(out)
    assert res["rccl_version"] >= 21800 and res["raw_ok"]
    assert res["bad_dtype"][0] != 0 and "dtype" in res["bad_dtype"][1]
    assert res["graphs"] == 1 and res["graph_dp"] == graph_form
    # per step: 4 BatchNorm layers x (forward + backward) SyncBN exchanges + ONE flat gradient all-reduce; with the peer exchange the two
    # encoder layers (BatchNorm + AvgPool: the one-launch kernel) exchange inside their kernels -- the decoder's of this small model keep
    # the un-folded Upsample, which that kernel does not take
    assert res["collectives_per_step"] == (9 if syncbn == "rccl" else 5)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.manual_seed(100)
        ref = get_trainer_dynamic(_args("mse"))
    for step in range(5):
        ref.train(synthetic_batch(3, 32, 32, seed=40 + step, brain=True), keep_predictions=False)
    np.testing.assert_allclose(res["loss"], ref.losses["loss_ae"].floats(), rtol=2e-5)
    for k, v in ref.model.state_dict().items():
        a, b = res["sd"][k].double(), v.cpu().double()
        if "num_batches" in k:
            assert int(a) == int(b)
            continue
        diff = (a - b).abs()
        assert float(diff.max()) <= 5 * 2 * 1e-3 + 1e-6, k
        assert float((diff > 1e-4 + 1e-3 * b.abs()).double().mean()) <= 0.03, k


# ---- first contact with a real multi-GPU node: TWO ranks on DISTINCT devices over RCCL -----------------------------------------------
def _rccl2_worker(rank, port, B, out, graph_form, steps, syncbn="rccl"):
    os.environ.update(RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      AESR_DP_GRAPH=graph_form or "segments", AESR_COMM_INIT_TIMEOUT="90", AESR_STEP_TIMEOUT="120", AESR_SYNCBN=syncbn)
    os.environ.pop("AESR_DIST_BACKEND", None)
    os.environ.pop("AESR_FORCE_DP", None)
    import warnings
    warnings.simplefilter("ignore")
    from superresolution_aniso_mri_amd.data_synth import shard_batch, synthetic_batch
    from superresolution_aniso_mri_amd.kwatsch.get_trainer import get_trainer_dynamic
    from superresolution_aniso_mri_amd.parallel import DataParallelContext
    torch.cuda.set_device(rank)
    dev = "cuda:%d" % rank
    dp = DataParallelContext(device=dev)
    try:
        assert dp.data_backend == "rccl"
        torch.manual_seed(100 + rank)                       # ranks start from different weights; attach() broadcasts rank 0's
        tr = get_trainer_dynamic(dict(_args("mse"), device=dev))
        dp.attach(tr)                                       # first collective: the checked first contact of parallel._first_contact
        dp.set_batch(B)
        if graph_form:
            tr.enable_step_graph(eager_steps=1)
        for step in range(steps):
            tr.train(shard_batch(synthetic_batch(B, 32, 32, seed=40 + step, brain=True), rank, 2), keep_predictions=False)
        dp.synchronize()
        from superresolution_aniso_mri_amd import _hip
        _hip.check_device_watchdogs("rank %d" % rank)
        losses = [dp.reduce_scalar(v) for v in tr.losses["loss_ae"].floats()]
        if rank == 0:
            torch.save({"sd": {k: v.cpu() for k, v in tr.model.state_dict().items()}, "loss": losses,
                        "graph_dp": getattr(tr, "_graph_dp", None), "ncoll": dp.n_collectives}, out)
        dp.barrier()
    finally:
        dp.shutdown()


@pytest.mark.parametrize("syncbn", ["rccl", "p2p"])
@pytest.mark.parametrize("graph_form", [None, "segments", "whole"])
@pytest.mark.parametrize("B", [4, 3])
def test_two_gpus_rccl_equal_single_process(tmp_path, graph_form, B, syncbn):
    """Needs two GPUs (skipped on the one-GPU build box; the driver's 8-GPU node runs it): two ranks on distinct devices through
    aesr_comm_init, 5 steps host-launched / as graph segments between eager RCCL collectives / as one graph with the collectives as
    nodes, even (2 + 2) and uneven (1 + 2 triplets) shards; every form must reproduce the single-process run."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import warnings
    from superresolution_aniso_mri_amd.data_synth import synthetic_batch
    from superresolution_aniso_mri_amd.kwatsch.get_trainer import get_trainer_dynamic
    out = str(tmp_path / "rccl2.pt")
    steps = 5
    try:
        mp.spawn(_rccl2_worker, args=(_free_port(), B, out, graph_form, steps, syncbn), nprocs=2, join=True)
    except Exception as e:                                  # noqa: BLE001
        # the opt-in form (multi-rank RCCL collectives captured as graph nodes) has never met two devices before this run.  Only a
        # REFUSAL to capture (RCCL / HIP saying the operation is not permitted or not supported while a stream is capturing) is an
        # expected finding about that form; a deadline abort, a GPU fault or a wrong number inside the worker fails the test, so
        # that its cause is found from this evidence
        msg = str(e).lower()
        refused = any(k in msg for k in ("operation not permitted when stream is capturing", "streamcaptureunsupported",
                                         "capture unsupported", "not supported during capture", "hiperrorstreamcapture"))
        if graph_form == "whole" and refused and "assert" not in msg and "did not finish within" not in msg:
            pytest.xfail("AESR_DP_GRAPH=whole on two devices: capture refused: %s" % (str(e)[-400:],))
        raise
    res = torch.load(out)
    assert res["graph_dp"] == graph_form or graph_form is None
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.manual_seed(100)
        ref = get_trainer_dynamic(_args("mse"))
    for step in range(steps):
        ref.train(synthetic_batch(B, 32, 32, seed=40 + step, brain=True), keep_predictions=False)
    np.testing.assert_allclose(res["loss"], ref.losses["loss_ae"].floats(), rtol=2e-5)
    for k, v in ref.model.state_dict().items():
        a, b = res["sd"][k].double(), v.cpu().double()
        if "num_batches" in k:
            assert int(a) == int(b)
            continue
        diff = (a - b).abs()
        assert float(diff.max()) <= steps * 2 * 1e-3 + 1e-6, k
        assert float((diff > 1e-4 + 1e-3 * b.abs()).double().mean()) <= 0.03, k
