"""``AEBaseTrainer``: constructor contract and plain-AE step of the reference's kwatsch/trainer_ae.py:16-109 on the
HIP engine.  Optimiser = fused-kernel Adam with the reference's hyper-parameters (:28-30); device-agnostic
construction (the reference's ``torch.cuda.FloatTensor`` at :51 made CPU construction impossible)."""
from collections import defaultdict

import torch

from .. import engine, ops
from .base_trainer import BaseTrainer, LossLog


class AEBaseTrainer(BaseTrainer):

    def __init__(self, args, ae, max_grad_norm=0, model_file=None, eval_mode=False, **kwargs):
        super(AEBaseTrainer, self).__init__()
        self.args = args
        self.model = ae
        self.model_sr = kwargs.get("model_sr", None)
        self.eval_model = eval_mode
        self.model_file = model_file
        self.do_chunk = False
        self.eval_fixed_coeff = True
        self.opt_ae = self._make_optimizer(args)
        self._init_scheduler()
        self.losses, self.losses_test = defaultdict(LossLog), defaultdict(LossLog)
        self.loss_iters = list()
        self.mean_losses, self.mean_losses_test = defaultdict(list), defaultdict(list)
        self.train_predictions, self.test_predictions = None, None
        self._iters = 1
        self.max_grad_norm = max_grad_norm            # accepted, never applied by the ae / ae_combined trainers
        self.use_multiple_gpu = False                 # the reference's "loss on cuda:1" trick is replaced by RCCL data parallel
        self.alpha05 = torch.tensor([0.5], dtype=torch.float32, device=args["device"])[:, None, None, None]
        self._init_laploss()
        self._init_percept_loss()
        self.determine_image_mix_loss_func()
        self.ssim_criterion = None
        self.epoch = 0
        self.init_weight_annealing(self.args.get("epochs", 1))
        if self.args.get("use_ssim_loss"):
            raise NotImplementedError("ERROR - SSIM as loss is disabled (as in the reference)")
        self.dp = None                                # set by parallel.DataParallelContext.attach()
        self._graph_enabled = False
        if model_file is not None:
            self.model_file = model_file
            self.load(model_file)
        if self.model_sr is not None and kwargs.get("model_file_sr", None) is not None:
            self.model_file_sr = kwargs.get("model_file_sr")
            self.load_caisr(self.model_file_sr)

    def _make_optimizer(self, args):
        params = list(self.model.parameters())
        momentum = args.get("momentum", 0.9)
        kw = dict(lr=args.get("lr", 1e-5), weight_decay=args.get("weight_decay", 0.0), betas=(momentum, 0.999))
        if params and params[0].is_cuda and not self.eval_model:
            return ops.HipAdam(params, on_step=self.model.mark_weights_dirty, **kw)
        # eval-only trainers / host-logic tests never step; a stock Adam keeps the checkpoint format
        return torch.optim.Adam(params, **kw)

    # ---- one optimisation step on an already built loss ----------------------------------------------------------
    def _backward_and_step(self, loss, eval_mode):
        self.opt_ae.zero_grad()
        if not eval_mode:
            one = self.__dict__.get("_one")
            if one is None or one.device != loss.device:
                one = self._one = torch.ones((), dtype=torch.float32, device=loss.device)      # seed of backward(): no ones_like fill per step
            # the weight-gradient slabs of the decoder and the encoder pass are summed by ONE launch at the end of the sweep
            with engine.deferred_wgrad_reductions():
                if self.dp is not None and self.dp.active:
                    (loss * self.dp.weight).backward(gradient=one)      # SUM over ranks of w_r * grad(loss_r) == grad of the global mean
                else:
                    loss.backward(gradient=one)
            if self.dp is not None and self.dp.active:
                self.dp.allreduce_gradients(self.opt_ae)
            self.opt_ae.step()
        elif self.dp is not None and self.dp.active:
            self.dp.step_fence()                 # no gradient all-reduce in this step: keep the ranks in step for the peer exchange
        if self.opt_sched_ae is not None:
            self.opt_sched_ae.step()
        self._poll_watchdogs()

    # ---- HIP-graph capture of the whole step (forward, backward, Adam) ---------------------------------------------
    def enable_step_graph(self, eager_steps=3, dp_segments=False, dp_mode=None):
        """Replay the training step from ONE captured HIP graph (no per-kernel host launches).  The first ``eager_steps``
        calls still run eagerly (allocator / plan caches and, under data parallel, RCCL's lazily built channels warm up), then
        the step is captured once per input signature.  Under data parallel the collectives are handled as the data plane allows
        (``dp_mode`` None = pick that form):
          "whole"    -- library-owned RCCL communicator: the collectives are enqueued on the capturing stream and become nodes
                        of the one step graph (capture mode "thread_local": RCCL's own proxy thread keeps running meanwhile);
          "segments" -- host-staged gloo data plane (CPU tests, several ranks rehearsed on one GPU): a chain of graphs cut at the
                        collectives, which stay eager (parallel.SegmentedStepGraph)."""
        dp_active = self.dp is not None and self.dp.active
        if dp_mode is None and (dp_segments or dp_active):
            dp_mode = self.dp.graph_mode if dp_active else "segments"
        if dp_mode not in (None, "whole", "segments"):
            raise ValueError("dp_mode must be None, 'whole' or 'segments', got %r" % (dp_mode,))
        if dp_active and dp_mode == "whole" and self.dp.data_backend != "rccl":
            raise ValueError("dp_mode='whole' needs the RCCL data plane (collectives of the %s data plane cannot be graph nodes)"
                             % self.dp.data_backend)
        self._graph_enabled = True
        self._graph_dp = dp_mode
        self._graph_eager_left = int(eager_steps)
        self._graphs = {}

    def _graph_ok(self, keep_predictions, eval_mode):
        if not getattr(self, "_graph_enabled", False) or keep_predictions or eval_mode:
            return False
        if self.opt_sched_ae is not None or self.args.get("get_masks"):
            return False                       # per-step learning rates / masks are not captured
        if self.dp is not None and self.dp.active and not getattr(self, "_graph_dp", False):
            return False                       # collectives: only with an explicit enable_step_graph(dp_mode=...)
        if self._graph_eager_left > 0:
            self._graph_eager_left -= 1
            return False
        return True

    def _train_graphed(self, dev_batch):
        keys = [k for k in ("image", "slice_between", "alpha_from", "alpha_to") if k in dev_batch]
        sig = tuple((k, tuple(dev_batch[k].shape)) for k in keys)
        g = self._graphs.get(sig)
        if g is None:
            if hasattr(self.model, "ensure_bn_barriers"):
                self.model.ensure_bn_barriers()         # state the captured kernels own: born outside the graph's memory pool
            # the weight-preparation launch must be a node of EVERY captured step: an eager pass since the last optimizer step (a
            # validate() right before the capture) leaves the packed operands fresh, the capture would then record no preparation at all and
            # every replay would run on the filters of the capture step
            self._replayed()
            # a "_persistent" batch (data_device.TripletAugmenter: one output buffer that every batch is written into) IS the static input:
            # later batches arrive at the same addresses and nothing is copied
            adopt = bool(dev_batch.get("_persistent"))
            static = {k: (dev_batch[k] if adopt else dev_batch[k].clone()) for k in keys}
            from ..networks.acai_vanilla import _joined_view
            joined = "image" in static and "slice_between" in static and _joined_view([static["image"], static["slice_between"]]) is not None
            if not joined and "image" in static and "slice_between" in static and static["image"].shape[1:] == static["slice_between"].shape[1:]:
                # the two image sub-batches back to back in ONE buffer: the encoder pass reads [image | slice_between] as it lies (no
                # concatenation node, networks/acai_vanilla._joined_view)
                n1 = static["image"].shape[0]
                both = torch.cat([static["image"], static["slice_between"]], dim=0).contiguous()
                static["image"], static["slice_between"] = both[:n1], both[n1:]
            sink = {}
            dp_active = self.dp is not None and self.dp.active
            if dp_active and self._graph_dp == "segments":
                from ..parallel import SegmentedStepGraph
                graph = SegmentedStepGraph()
                self.dp.segments = graph
                self._capture_sink = sink
                try:
                    graph.capture(lambda: self._step_core(static, False))      # runs the step once (segment by segment)
                finally:
                    self._capture_sink = None
                    self.dp.segments = None
                self._graphs[sig] = (graph, static, sink)
                self._replayed()
                self._log_sink(sink)
                return
            graph = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(graph, capture_error_mode="thread_local" if dp_active else "global"):
                self._capture_sink = sink
                try:
                    self._step_core(static, False)
                finally:
                    self._capture_sink = None
            g = self._graphs[sig] = (graph, static, sink)
        graph, static, sink = g
        for k in keys:
            if static[k].data_ptr() != dev_batch[k].data_ptr():
                static[k].copy_(dev_batch[k])
        graph.replay()
        self._replayed()
        self._log_sink(sink)
        self._poll_watchdogs()

    WATCHDOG_EVERY = 512        # steps between two polls of the kernel-side watchdogs (a poll reads a device symbol: it syncs)

    def _poll_watchdogs(self):
        """Round-4 advice: a grid barrier / peer wait that gave up lets the step run on garbage; validate(), the epoch log and the
        checkpoints refuse to go on, but an epoch can be thousands of steps long -- so the step itself looks every WATCHDOG_EVERY
        iterations (one device sync per ~1 s of training at 2 ms per step)."""
        if (self._iters % self.WATCHDOG_EVERY) == 0 and torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
            from .base_trainer import _check_watchdogs
            self._bounded_sync()        # data parallel: the symbol read below syncs the device -- wait with the step deadline first
            _check_watchdogs(self, "train step %d" % self._iters)

    def _replayed(self):
        """A replay rewrote the parameters, the BatchNorm running statistics and (one step behind) the packed operands through raw
        pointers: no tensor version moved and no Python ran, so every host-side cache keyed on them is stale -- the eval-mode BatchNorm
        scale / shift, the packed filters an eager validate() / encode() / decode() would reuse.  Host counters only."""
        if hasattr(self.model, "mark_weights_dirty"):
            self.model.mark_weights_dirty()

    def _log_sink(self, sink):
        """Log the scalars a replayed step left in its static buffers: ONE gather kernel for all of them (the buffers are
        overwritten by the next replay, so the values must be copied out; a clone per scalar costs a launch each)."""
        keys = list(sink.keys())
        vals = torch.stack([sink[k].detach().reshape(()) for k in keys])
        for i, k in enumerate(keys):
            self.losses[k].append(vals[i])

    def train(self, batch_item, keep_predictions=True, eval_mode=False):
        """Plain ``ae`` step (reference :71-109): reconstruction loss only; latent loss and the 0.5-mix are logged."""
        x = self._to_device(batch_item["image"])
        self._set_mode(not eval_mode)
        self._iters += 1
        self._note_batch(batch_item, "train")
        if hasattr(self.model, "prepare_weights"):
            self.model.prepare_weights()
        if self.dp is not None and self.dp.active:
            self.dp.begin_step()
        z = self.model.encode(x)
        out = self.model.decode(z)
        loss_ae = self.get_loss(x, out, is_test=False)["loss_ae"]
        a_from, a_to = self._mix_coefficients(batch_item, x.shape[0] // 2)
        lat = self.get_latent_loss(reference=self._to_device(batch_item["slice_between"]), alpha_from=a_from, alpha_to=a_to,
                                   z=z.detach(), no_grad=True)
        self._set_mode(not eval_mode)
        self._backward_and_step(loss_ae, eval_mode)
        self._log("loss_ae", loss_ae)
        self._log("loss_latent_1", lat["loss_latent"])
        if keep_predictions:
            # reference :100-102: get_latent_loss(no_grad=True) left the model in EVAL mode (BaseTrainer.encode(eval=True)) and
            # nothing switches it back before _get_mixup_image: the logged 0.5-mix is decoded with the running statistics and
            # does not update them
            self._set_mode(False)
            mix = self._get_mixup_image(z=z.detach(), alpha_from=a_from, alpha_to=a_to, is_test=True)
            self._set_mode(not eval_mode)
            s = mix["slice_inbetween_mix"].detach().cpu()
            self.train_predictions = {"z_mix": lat["z_mix"].detach().cpu(), "pred_alphas": torch.tensor([0.5]),
                                      "slice_inbetween_mix": s, "slice_inbetween_05": s, "reconstruction": out.detach().cpu()}
