"""Data parallelism for the ae_combined step: one process per GPU; the device collectives run on an RCCL communicator (xGMI)
that the C-ABI library owns (csrc/comm.hip), ``torch.distributed`` (gloo) is the host-side control plane only.
New functionality -- the reference has no distributed path.

The unit that shards is the *triplet* (from, to, between).  Per step there are exactly these exchanges:
  * gradients   ONE all-reduce(SUM) of the flat fp32 gradient buffer (1.78 MB for the ACDC model: latency-bound,
                so a single flat collective instead of per-tensor buckets);
  * SyncBN      per BatchNorm call an all-reduce(SUM) of [G][2][C] fp64 partial sums (+ element counts) forward and
                [G][2][C] backward -- statistics are those of the GLOBAL sub-batch, as in the single-process reference;
  * logging     loss scalars are all-reduced only when they are read.
Uneven shards (12 triplets over 8 ranks = 2,2,2,2,1,1,1,1) stay exact: rank r back-propagates w_r * loss_r with
w_r = B_r / B_global, so SUM over ranks is the gradient of the global mean."""
import os

import torch
import torch.distributed as dist


class SegmentedStepGraph(object):
    """The data-parallel step as a CHAIN of HIP graphs cut at every collective, which stays an EAGER call between two replays: the
    default form (``AESR_DP_GRAPH=segments``) for the RCCL data plane -- plain ncclAllReduce enqueues on the stream, RCCL's most
    travelled path -- and the only one for a HOST-staged data plane (gloo: CPU tests and several ranks rehearsed on one GPU), whose
    collectives cannot be graph nodes.  ``AESR_DP_GRAPH=whole`` captures the RCCL collectives as nodes of ONE step graph instead
    (``dp_mode="whole"``: ~10 graph launches and 9 eager enqueues fewer per step; until a multi-GPU run has exercised captured
    multi-rank collectives it is opt-in).

    During the capture step every ``cut(fn)`` ends the running capture, replays that segment (its results are needed now), runs
    the collective ``fn`` eagerly and begins the next segment.  Later steps replay segment i, then call collective i on the very
    tensor it was recorded with (kept alive here; all segments share one memory pool, so addresses repeat)."""

    def __init__(self):
        self.graphs, self.collectives = [], []
        self._cur, self._pool = None, None
        self.capturing = False

    def _begin(self):
        g = torch.cuda.CUDAGraph()
        if self._pool is None:
            self._pool = torch.cuda.graph_pool_handle()          # one memory pool shared by all segments
        g.capture_begin(pool=self._pool, capture_error_mode="relaxed")
        self._cur = g

    def _end(self):
        self._cur.capture_end()
        self._cur.replay()
        self.graphs.append(self._cur)
        self._cur = None

    def capture(self, fn):
        """Run ``fn()`` (the whole step) once, recording it as graph segments; collectives must go through ``cut``."""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            self.capturing = True
            try:
                self._begin()
                fn()
                self._end()
            finally:
                self.capturing = False
                if self._cur is not None:          # an exception inside a segment: close the capture before propagating
                    try:
                        self._cur.capture_end()
                    except Exception:              # noqa: BLE001
                        pass
                    self._cur = None
        torch.cuda.current_stream().wait_stream(side)

    def cut(self, fn):
        self._end()
        fn()
        self.collectives.append(fn)
        self._begin()

    def replay(self):
        for i, g in enumerate(self.graphs):
            g.replay()
            if i < len(self.collectives):
                self.collectives[i]()


class PeerExchange(object):
    """The one-shot SyncBN exchange (``AESR_SYNCBN=p2p``; csrc/p2p.hip, csrc/bn_fused.hip): every rank owns a small region of fine-grained
    device memory that all ranks map through IPC handles (handed round on the gloo control plane); a BatchNorm call's partial sums are
    then written straight into every peer's region by the ONE kernel that also computes the statistics and normalises -- no RCCL launch and
    no graph cut per BatchNorm call (8 per step), only the gradient all-reduce is left.  SURVEY section 5: "one-shot ... direct P2P write
    over the 7 links, which beats a ring at this size".  OPT-IN until a multi-GPU box has run it."""

    def __init__(self, rank, world, device):
        import ctypes
        from ._hip import P2P_HANDLE_BYTES, check, lib
        if world > 8:
            raise RuntimeError("the peer exchange is built for one node (<= 8 ranks), got %d" % world)
        self.rank, self.world = rank, world
        self.nbytes = int(lib.aesr_p2p_region_bytes(world))
        own = ctypes.c_void_p()
        check(lib.aesr_p2p_alloc(self.nbytes, ctypes.byref(own)), "aesr_p2p_alloc")
        self.own = own
        self.opened = []
        regions = [None] * world
        regions[rank] = own.value
        if world > 1:
            h = ctypes.create_string_buffer(P2P_HANDLE_BYTES)
            check(lib.aesr_p2p_get_handle(own, h), "aesr_p2p_get_handle")
            handles = [None] * world
            dist.all_gather_object(handles, h.raw)              # 64 bytes per rank over the gloo/TCP control plane
            for r in range(world):
                if r == rank:
                    continue
                peer = ctypes.c_void_p()
                check(lib.aesr_p2p_open(handles[r], ctypes.byref(peer)), "aesr_p2p_open(rank %d)" % r)
                regions[r] = peer.value
                self.opened.append(peer)
            dist.barrier()                                      # nobody writes before everybody has mapped everything
        self.peers = (ctypes.c_void_p * 8)(*(regions + [None] * (8 - world)))
        self.gen = torch.zeros(1, dtype=torch.int32, device=device)       # advanced once per step by aesr_p2p_tick
        self.slot = 0
        self.nscale = 1.0            # largest shard / this rank's shard (uneven shards): every rank must take the same kernel decision

    def begin_step(self):
        """Once per training step, before its first BatchNorm call: advance the generation (a kernel: captured like the rest of the step)
        and start numbering the step's BatchNorm calls again."""
        from ._hip import check, lib, ptr, stream
        check(lib.aesr_p2p_tick(ptr(self.gen), stream()), "aesr_p2p_tick")
        self.slot = 0

    def next_slot(self):
        from ._hip import P2P_SLOTS
        s = self.slot
        if s >= P2P_SLOTS:
            raise RuntimeError("more than %d BatchNorm exchanges in one step" % P2P_SLOTS)
        self.slot += 1
        return s

    def close(self):
        from ._hip import lib
        for peer in self.opened:
            lib.aesr_p2p_close(peer)
        self.opened = []
        if self.own is not None:
            lib.aesr_p2p_free(self.own)
            self.own = None


class DataParallelContext(object):
    """Control plane: a ``torch.distributed`` **gloo** group (rendezvous, barriers, host scalars, the RCCL unique id).
    Data plane on GPUs: an RCCL communicator owned behind the C ABI (``aesr_comm_*``, csrc/comm.hip) whose collectives are
    plain enqueues on the current stream -- capturable into the step graph, no ProcessGroupNCCL and therefore no watchdog thread
    polling events of a capturing stream.  ``AESR_DIST_BACKEND=gloo`` (CPU tests, several ranks rehearsed on ONE GPU, where RCCL
    refuses duplicate devices) stages device tensors through the host instead."""

    def __init__(self, backend=None, device=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = device
        backend = backend or os.environ.get("AESR_DIST_BACKEND")
        if backend == "nccl":
            backend = "rccl"                # "nccl" used to mean ProcessGroupNCCL here; the data plane is the library's own comm now
        if backend not in (None, "gloo", "rccl"):
            raise ValueError("data-parallel backend must be 'rccl' or 'gloo', got %r" % (backend,))
        self.data_backend = backend or ("rccl" if torch.cuda.is_available() else "gloo")
        # rehearsal switch: treat a single process as "data parallel" (collectives over a group of one) so that the RCCL call
        # pattern, incl. the captured step graph, can be exercised on a one-GPU box
        self._force = os.environ.get("AESR_FORCE_DP", "0") == "1"
        if (self.world > 1 or self._force) and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            dist.init_process_group(backend="gloo", rank=self.rank, world_size=self.world)
        self.comm = None                # aesr_comm handle (data_backend == "rccl"), created on first use with the device current
        self.weight = 1.0
        self.global_B = None
        self.segments = None            # a SegmentedStepGraph while a step is being captured (gloo data plane only)
        self.n_collectives = 0          # collectives issued so far (tests / DESIGN accounting)
        # SyncBN exchange: "rccl" (default: an all-reduce per BatchNorm call on the data plane) or "p2p" (opt-in: one-shot writes into
        # peer-mapped regions inside the BatchNorm kernels, PeerExchange)
        self.syncbn = os.environ.get("AESR_SYNCBN", "rccl")
        if self.syncbn not in ("rccl", "p2p"):
            raise ValueError("AESR_SYNCBN must be 'rccl' or 'p2p', got %r" % (self.syncbn,))
        self.p2p = None
        self._env_saved = {}            # process environment this context changed (rehearsal settings): restored by shutdown()
        if self.world > 1 and os.environ.get("AESR_SINGLE_DEVICE") == "1":
            # several ranks REHEARSED on one device: the one-launch BatchNorm kernels of all ranks must be resident TOGETHER (their grid
            # barriers and peer waits would otherwise spin until they give up): 256 / world workgroups each, or the three-launch form.
            # The library reads both switches per call, so they live in the environment -- for the life of THIS context only (round-5
            # advice: a later single-process trainer in the same process must not inherit the shrunken grid or the disabled kernel)
            if self.world <= 4:
                if "AESR_BN_FUSED_NB" not in os.environ:
                    self._set_env("AESR_BN_FUSED_NB", "128" if self.world == 2 else "64")
            elif os.environ.get("AESR_BN_FUSED", "1") != "0":
                if self.syncbn == "p2p":
                    raise ValueError("AESR_SYNCBN=p2p with %d ranks on ONE device: their one-launch BatchNorm kernels cannot all be resident" % self.world)
                self._set_env("AESR_BN_FUSED", "0")

    def _set_env(self, key, value):
        self._env_saved.setdefault(key, os.environ.get(key))
        os.environ[key] = value

    def _restore_env(self):
        for key, old in self._env_saved.items():
            if old is None:
                os.environ.pop(key, None)
            else:
                os.environ[key] = old
        self._env_saved = {}

    @property
    def active(self):
        return self.world > 1 or self._force

    @property
    def graph_mode(self):
        """How a captured step handles the collectives: "segments" (eager calls between graph segments; default) or "whole" (RCCL
        calls are nodes of one step graph; ``AESR_DP_GRAPH=whole``, RCCL data plane only)."""
        want = os.environ.get("AESR_DP_GRAPH", "segments")
        if want not in ("segments", "whole"):
            raise ValueError("AESR_DP_GRAPH must be 'segments' or 'whole', got %r" % (want,))
        return "whole" if (want == "whole" and self.data_backend == "rccl") else "segments"

    # ---- the library-owned RCCL communicator -------------------------------------------------------------------
    def ensure_comm(self):
        if self.comm is not None or self.data_backend != "rccl" or not self.active:
            return self.comm
        import ctypes
        from ._hip import COMM_ID_BYTES, check, lib
        uid = ctypes.create_string_buffer(COMM_ID_BYTES)
        if self.rank == 0:
            check(lib.aesr_comm_unique_id(uid), "aesr_comm_unique_id")
        box = [uid.raw]
        if self.world > 1:
            dist.broadcast_object_list(box, src=0)           # 128 bytes over the gloo/TCP control plane
        handle = ctypes.c_void_p()
        # collective over all ranks; the CURRENT device becomes this rank's device (callers set it before the first collective)
        check(lib.aesr_comm_init(box[0], self.world, self.rank, ctypes.byref(handle)), "aesr_comm_init")
        self.comm = handle
        self._first_contact()
        return self.comm

    def _first_contact(self):
        """The communicator's first collective, checked: every rank contributes rank + 1 (fp32 and fp64), the sum must be
        world (world + 1) / 2 on every rank and must arrive within the deadline -- a dead peer, a wrong device binding or a broken
        fabric path shows up HERE with a message, not as a hang in step 1."""
        n = self.world
        a = torch.full((1024,), float(self.rank + 1), dtype=torch.float32, device=self.device)
        b = torch.full((64,), float(self.rank + 1), dtype=torch.float64, device=self.device)
        self._rccl(a, "sum")
        self._rccl(b, "sum")
        self.synchronize(float(os.environ.get("AESR_COMM_INIT_TIMEOUT", "120")), "the first all-reduce of the RCCL communicator")
        want = n * (n + 1) / 2.0
        if float(a.min()) != want or float(a.max()) != want or float(b.min()) != want or float(b.max()) != want:
            raise RuntimeError("RCCL first contact: all-reduce over %d ranks returned [%g, %g] / [%g, %g], expected %g" % (
                n, float(a.min()), float(a.max()), float(b.min()), float(b.max()), want))

    def synchronize(self, timeout_s=None, what="the data-parallel step"):
        """``torch.cuda.synchronize()`` with a deadline: the library's communicator has no watchdog thread (csrc/comm.hip), so a dead or
        late peer would otherwise hang this rank forever.  On timeout the communicator is aborted (ncclCommAbort) and RuntimeError is
        raised; callers exit non-zero."""
        import time
        if timeout_s is None:
            timeout_s = float(os.environ.get("AESR_STEP_TIMEOUT", "300"))
        if self.comm is None or not torch.cuda.is_available():
            torch.cuda.synchronize() if torch.cuda.is_available() else None
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        t0 = time.perf_counter()
        while not ev.query():
            if time.perf_counter() - t0 > timeout_s:
                from ._hip import lib
                try:
                    lib.aesr_comm_abort(self.comm)
                finally:
                    self.comm = None
                raise RuntimeError("rank %d: %s did not finish within %.0f s (a peer is dead or late); RCCL communicator aborted" % (
                    self.rank, what, timeout_s))
            time.sleep(0.0005 if time.perf_counter() - t0 < 0.05 else 0.01)

    def _rccl(self, t, op):
        from ._hip import COMM_F32, COMM_F64, COMM_MAX, COMM_SUM, check, lib, ptr, stream
        if not t.is_cuda or not t.is_contiguous():
            raise RuntimeError("RCCL collectives take contiguous device tensors (got %s, contiguous=%s)" % (t.device, t.is_contiguous()))
        if t.dtype == torch.float32:
            dt = COMM_F32
        elif t.dtype == torch.float64:
            dt = COMM_F64
        else:
            raise RuntimeError("RCCL all-reduce of %s is not wired" % t.dtype)
        check(lib.aesr_comm_allreduce(self.ensure_comm(), ptr(t), t.numel(), dt, COMM_MAX if op == "max" else COMM_SUM, stream()),
              "aesr_comm_allreduce")

    def _all_reduce(self, t, op="sum"):
        """In-place all-reduce.  Device tensors go through the library's RCCL communicator on the current stream (data backend
        "rccl"); with the gloo data backend (CPU tests, one-GPU rehearsals) they are staged through the host.  Host tensors always
        use gloo."""
        self.n_collectives += 1
        if t.is_cuda and self.data_backend == "rccl":
            if self.segments is not None and self.segments.capturing:
                self.segments.cut(lambda: self._rccl(t, op))      # eager enqueue between two graph segments, replayed on this tensor
            else:
                self._rccl(t, op)                                 # eager step, or a node of the whole-step graph being captured
            return
        rop = dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM

        def run():
            if t.is_cuda:
                h = t.detach().cpu()
                dist.all_reduce(h, op=rop)
                t.copy_(h)
            else:
                dist.all_reduce(t, op=rop)

        if self.segments is not None and self.segments.capturing:
            self.segments.cut(run)          # eager, between two graph segments; replayed on this same tensor later
        else:
            run()

    def shard_range(self, B):
        return (B * self.rank) // self.world, (B * (self.rank + 1)) // self.world

    def set_batch(self, B_global):
        lo, hi = self.shard_range(B_global)
        self.global_B = B_global
        self.weight = float(hi - lo) / float(B_global)
        self._update_model()
        return lo, hi

    def ensure_p2p(self):
        if self.p2p is None and self.syncbn == "p2p" and self.active and torch.cuda.is_available():
            torch.cuda.synchronize()
            self.p2p = PeerExchange(self.rank, self.world, self.device)
        return self.p2p

    def begin_step(self):
        """Top of every training step (trainers call it): the peer exchange advances its generation."""
        if self.p2p is not None:
            self.p2p.begin_step()

    def step_fence(self):
        """A step WITHOUT a gradient all-reduce (``train(..., eval_mode=True)``) under the peer exchange: the exchange re-uses its slots
        on the promise that every rank has consumed step t before any rank writes step t + 1, which the gradient all-reduce keeps -- here a
        4-byte all-reduce stands in for it."""
        if self.p2p is not None:
            buf = self.__dict__.get("_fence_buf")
            if buf is None:
                buf = self._fence_buf = torch.zeros(1, dtype=torch.float32, device=self.device)
            self._all_reduce(buf)

    def _update_model(self):
        tr = getattr(self, "trainer", None)
        if tr is not None and hasattr(tr.model, "set_sync_bn"):
            p2p = self.ensure_p2p()
            if p2p is not None and self.global_B:
                lo, hi = self.shard_range(self.global_B)
                largest = -(-self.global_B // self.world)
                p2p.nscale = float(largest) / float(max(1, hi - lo))
            tr.model.set_sync_bn(self.sync_bn, 1.0 / self.weight if self.weight > 0 else 1.0, p2p=p2p)

    # ---- hooks -----------------------------------------------------------------------------------------------
    def sync_bn(self, sums):
        """All-reduce BatchNorm partial sums across ranks, in place (element counts are known on the host: every
        sub-batch of a step scales by the same B_global / B_local, see ``count_scale``)."""
        if not self.active:
            return
        self._all_reduce(sums)

    def allreduce_gradients(self, opt):
        if not self.active:
            return
        flat = getattr(opt, "flat_g", None)
        if flat is not None:
            if hasattr(opt, "_settle_unwritten"):
                opt._settle_unwritten()          # lazy zero_grad: parameters no pass differentiated hold zeros before the sum
            self._all_reduce(flat)
            return
        grads = [p.grad for g in opt.param_groups for p in g["params"] if p.grad is not None]
        buf = torch.cat([g.reshape(-1) for g in grads])
        self._all_reduce(buf)
        off = 0
        for g in grads:
            g.copy_(buf[off:off + g.numel()].view_as(g))
            off += g.numel()

    def broadcast_parameters(self, model):
        if not self.active:
            return
        from ._hip import COMM_F32, check, lib, ptr, stream
        for t in list(model.parameters()) + list(model.buffers()):
            d = t.data
            if d.is_cuda and self.data_backend == "rccl":
                if not d.is_contiguous() or (d.numel() * d.element_size()) % 4 != 0:
                    raise RuntimeError("broadcast needs contiguous tensors of whole 32-bit words")
                # raw 32-bit words: fp32 parameters and the int64 num_batches_tracked counters alike
                check(lib.aesr_comm_broadcast(self.ensure_comm(), ptr(d), d.numel() * d.element_size() // 4, COMM_F32, 0, stream()),
                      "aesr_comm_broadcast")
            elif d.is_cuda:
                h = d.cpu()
                dist.broadcast(h, src=0)
                d.copy_(h)
            else:
                dist.broadcast(d, src=0)
        if hasattr(model, "mark_weights_dirty"):
            model.mark_weights_dirty()

    def attach(self, trainer):
        """Make ``trainer`` data parallel: SyncBN hooks, gradient all-reduce, identical initial parameters."""
        trainer.dp = self
        self.trainer = trainer
        self._update_model()
        self.broadcast_parameters(trainer.model)
        return trainer

    def reduce_scalar(self, v, weighted=True):
        if not self.active:
            return v
        t = torch.tensor([float(v)], dtype=torch.float64)          # host scalar on the control plane
        if weighted:
            t *= self.weight
        self._all_reduce(t)
        return float(t)

    def reduce_means(self, means, n_local):
        """{key: mean over this rank's shard} -> {key: mean over the global batch}: sum_r n_r * mean_r / sum_r n_r with ONE
        all-reduce (keys in sorted order; every rank must log the same keys).  Used when losses are READ (epoch logging, model
        selection), never inside the step."""
        if not self.active or not means:
            return means
        keys = sorted(means.keys())
        t = torch.tensor([means[k] * n_local for k in keys] + [n_local], dtype=torch.float64)
        self._all_reduce(t)
        tot = float(t[-1])
        vals = t[:-1].tolist()
        return {k: (v / tot if tot > 0 else float("nan")) for k, v in zip(keys, vals)}

    def barrier(self):
        if self.active and self.world > 1:
            dist.barrier()

    def shutdown(self):
        """Tear the communicator and the process group down (quiet exit under torch.distributed.run)."""
        self._restore_env()
        if self.p2p is not None:
            try:
                if torch.cuda.is_available():
                    torch.cuda.synchronize()
                if self.world > 1 and dist.is_initialized():
                    dist.barrier()              # nobody unmaps a region a peer may still write to
                self.p2p.close()
            except Exception:              # noqa: BLE001
                pass
            self.p2p = None
        if self.comm is not None:
            from ._hip import lib
            try:
                self.synchronize(float(os.environ.get("AESR_SHUTDOWN_TIMEOUT", "60")), "the work still queued at shutdown")
                if self.comm is not None:
                    lib.aesr_comm_destroy(self.comm)
            except Exception:              # noqa: BLE001  (synchronize() aborted the communicator: nothing left to destroy)
                pass
            self.comm = None
        if dist.is_available() and dist.is_initialized():
            try:
                dist.destroy_process_group()
            except Exception:              # noqa: BLE001
                pass

    def max_over_ranks(self, v):
        if not self.active:
            return v
        t = torch.tensor([float(v)], dtype=torch.float64)
        self._all_reduce(t, "max")
        return float(t)


def one_rank_collective_cost_us(device, reps=50):
    """Device time per collective of a data-parallel step's exchange pattern on a ONE-rank RCCL communicator: 8 SyncBN all-reduces
    ([2][2][64] fp64) and the flat gradient all-reduce (443 777 fp32), as they are enqueued on the stream.  No wire time -- with one
    rank RCCL copies in place -- so this is the enqueue / launch floor of the 9 collectives, the part of their cost that does not shrink
    with the batch (bench.py: projected 8-rank rate)."""
    import ctypes
    from ._hip import COMM_F32, COMM_F64, COMM_ID_BYTES, COMM_SUM, check, lib, ptr, stream
    uid = ctypes.create_string_buffer(COMM_ID_BYTES)
    check(lib.aesr_comm_unique_id(uid), "aesr_comm_unique_id")
    handle = ctypes.c_void_p()
    torch.cuda.set_device(device)
    check(lib.aesr_comm_init(uid.raw, 1, 0, ctypes.byref(handle)), "aesr_comm_init")
    try:
        small = [torch.zeros((2, 2, 64), dtype=torch.float64, device=device) for _ in range(8)]
        flat = torch.zeros(443777, dtype=torch.float32, device=device)

        def one_step():
            for t in small:
                check(lib.aesr_comm_allreduce(handle, ptr(t), t.numel(), COMM_F64, COMM_SUM, stream()), "aesr_comm_allreduce")
            check(lib.aesr_comm_allreduce(handle, ptr(flat), flat.numel(), COMM_F32, COMM_SUM, stream()), "aesr_comm_allreduce")
        for _ in range(5):
            one_step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            one_step()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / (9 * reps)
    finally:
        lib.aesr_comm_destroy(handle)
