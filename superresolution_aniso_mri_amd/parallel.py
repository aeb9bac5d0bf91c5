"""Data parallelism for the ae_combined step: one process per GPU, ``torch.distributed`` (backend "nccl" == RCCL over
xGMI on ROCm; "gloo" in CPU tests).  New functionality -- the reference has no distributed path.

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
    """The data-parallel step as a CHAIN of HIP graphs cut at every collective.

    Capturing an RCCL collective into a graph is not possible on this stack (the ProcessGroupNCCL watchdog queries an event
    while the stream is capturing and the process aborts: scripts/nccl_smoke.py), so the collectives stay eager: during the
    capture step every ``cut(fn)`` ends the running capture, replays that segment (its results are needed now), runs the
    collective ``fn`` eagerly and begins the next segment.  Later steps replay segment i, then call collective i on the very
    tensor it was recorded with (kept alive here; all segments share one memory pool, so addresses repeat).
    Capture mode is "relaxed": segments end on the autograd thread and the watchdog thread may query events meanwhile."""

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


class DataParallelContext(object):

    def __init__(self, backend=None, device=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = device
        if self.world > 1 and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            # AESR_DIST_BACKEND=gloo: rehearsal of the N-rank launch on a box with fewer GPUs than ranks (RCCL refuses two ranks
            # on one device); production = nccl (RCCL over xGMI)
            backend = backend or os.environ.get("AESR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world)
        self.weight = 1.0
        self.global_B = None
        self.segments = None            # a SegmentedStepGraph while a step is being captured
        # rehearsal switch: treat a single process as "data parallel" (collectives over a group of one) so that the
        # NCCL call pattern, incl. the segmented step graph, can be exercised on a one-GPU box
        self._force = os.environ.get("AESR_FORCE_DP", "0") == "1"
        if self._force and self.world == 1 and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
            dist.init_process_group(backend=backend, rank=0, world_size=1)

    @property
    def active(self):
        return self.world > 1 or self._force

    def _all_reduce(self, t, op=None):
        """In-place all-reduce; with the gloo backend (CPU tests, single-GPU rehearsals) device tensors are staged through
        the host, with nccl (= RCCL over xGMI) they are reduced in place on the device."""
        op = op or dist.ReduceOp.SUM

        def run():
            if t.is_cuda and dist.get_backend() == "gloo":
                h = t.detach().cpu()
                dist.all_reduce(h, op=op)
                t.copy_(h)
            else:
                dist.all_reduce(t, op=op)

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

    def _update_model(self):
        tr = getattr(self, "trainer", None)
        if tr is not None and hasattr(tr.model, "set_sync_bn"):
            tr.model.set_sync_bn(self.sync_bn, 1.0 / self.weight if self.weight > 0 else 1.0)

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
        for t in list(model.parameters()) + list(model.buffers()):
            if t.is_cuda and dist.get_backend() == "gloo":
                h = t.data.cpu()
                dist.broadcast(h, src=0)
                t.data.copy_(h)
            else:
                dist.broadcast(t.data, src=0)
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
        t = torch.as_tensor(v, dtype=torch.float64, device=self.device).clone().reshape(1)
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
        t = torch.tensor([means[k] * n_local for k in keys] + [n_local], dtype=torch.float64, device=self.device)
        self._all_reduce(t)
        tot = float(t[-1])
        vals = t[:-1].cpu().tolist()
        return {k: (v / tot if tot > 0 else float("nan")) for k, v in zip(keys, vals)}

    def barrier(self):
        if self.active:
            dist.barrier()

    def shutdown(self):
        """Tear the process group down (quiet exit under torch.distributed.run)."""
        if dist.is_available() and dist.is_initialized():
            try:
                dist.destroy_process_group()
            except Exception:              # noqa: BLE001
                pass

    def max_over_ranks(self, v):
        if not self.active:
            return v
        t = torch.tensor([float(v)], dtype=torch.float64, device=self.device)
        self._all_reduce(t, dist.ReduceOp.MAX)
        return float(t)
