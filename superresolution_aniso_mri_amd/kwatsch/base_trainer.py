"""Trainer base of the MI355X build: the drop-in surface of the reference's ``BaseTrainer``
(kwatsch/base_trainer.py:16-460) -- loss bookkeeping, ``get_loss`` / ``get_latent_loss``, eval wrappers
``encode`` / ``decode`` / ``predict``, latent mixing, validation, checkpoints, loss files -- rebuilt device-agnostic
on the HIP ops (the reference hard-codes 'cuda' and chunks 256x256 batches through the CPU to fit an 11 GB card;
on a 288 GB MI355X the chunking is dropped: eval-mode BatchNorm makes chunked == unchunked).

Scalars are logged without forcing a device sync per step: ``self.losses[key]`` holds 0-dim device tensors that are
converted to floats when read."""
import os
from collections import defaultdict

import numpy as np
import torch

from .. import ops


def _check_watchdogs(trainer, where):
    """Kernel-side protocol watchdogs (``_hip.check_device_watchdogs``), for trainers whose model lives on the GPU (host-logic tests build
    CPU trainers that never launch a kernel)."""
    model = getattr(trainer, "model", None)
    p = next(iter(model.parameters()), None) if model is not None else None
    if p is not None and p.is_cuda:
        from .._hip import check_device_watchdogs
        check_device_watchdogs(where)


class LossLog(list):
    """list of scalars; device tensors are turned into floats lazily (on read), so appending never syncs."""

    def _conv(self, i):
        v = list.__getitem__(self, i)
        if torch.is_tensor(v):
            v = float(v.detach().float().cpu())
            list.__setitem__(self, i, v)
        return v

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._conv(k) for k in range(*i.indices(len(self)))]
        return self._conv(i if i >= 0 else len(self) + i)

    def __iter__(self):
        return (self._conv(k) for k in range(len(self)))

    def floats(self):
        return [self._conv(k) for k in range(len(self))]


def optimizer_state_to_cpu(opt):
    """A CPU copy of ``opt.state_dict()`` for a checkpoint.  ``Optimizer.state_dict()`` hands out the SAME inner dicts as
    ``opt.state[p]``, so they must never be written to: replacing a moment tensor there would detach HipAdam's flat-buffer
    views from the live optimizer state and every later checkpoint would carry the moments of the first one."""
    sd = opt.state_dict()
    state = {pid: {k: (v.detach().cpu().clone().contiguous() if torch.is_tensor(v) else v) for k, v in st.items()}
             for pid, st in sd["state"].items()}
    groups = [dict(g, params=list(g["params"])) for g in sd["param_groups"]]
    return {"state": state, "param_groups": groups}


def _scalar(v):
    return v.detach() if torch.is_tensor(v) else float(v)


class BaseTrainer(object):

    # ---- construction helpers (called by AEBaseTrainer.__init__) --------------------------------------------
    def _init_scheduler(self):
        self.opt_sched_ae = None
        if self.args.get("use_lr_scheduler"):
            self.opt_sched_ae = torch.optim.lr_scheduler.CosineAnnealingLR(self.opt_ae, self.args["lr_iter_max"], eta_min=0,
                                                                           last_epoch=-1)

    def _wants_lpips(self):
        a = self.args
        return bool(a.get("use_percept_loss")) or a.get("image_mix_loss_func") == "perceptual" or \
            a.get("alpha_loss_func") == "perceptual"

    def _init_percept_loss(self):
        self.ae_loss_func = "perceptual" if self.args.get("use_percept_loss") else "mse"
        self.percept_criterion = None
        if not self.eval_model and self._wants_lpips():
            from ..lpips.perceptual import PerceptualLoss
            self.percept_criterion = PerceptualLoss(model="net-lin", net="vgg", use_gpu=str(self.args["device"]).startswith("cuda"),
                                                    gpu_ids=[0], vgg_weights=self.args.get("vgg_weights"),
                                                    device=self.args["device"])

    def _init_laploss(self):
        self.laploss = None
        if not self.eval_model and self.args.get("use_laploss"):          # reference :47-56
            from .lap_pyramid_loss import LapLoss
            self.laploss = LapLoss(channels=1, device=self.args["device"])

    def determine_image_mix_loss_func(self):
        f = self.args.get("image_mix_loss_func")
        self.image_mix_loss_func = f if f is not None else ("perceptual" if self.args.get("use_percept_loss") else "mse")

    def init_weight_annealing(self, epochs):
        # reference :456-459: sigmoid(linspace(-5,5,epochs)) * lambda, reversed
        x = np.linspace(-5, 5, epochs)
        self.loss_weights = (1.0 / (1.0 + np.exp(-x)) * self.args.get("ex_loss_weight1", 0.001))[::-1].copy()

    def init_weight_ramp(self, epochs):
        x = np.linspace(-2, 10, epochs)
        self.loss_weights = 1.0 / (1.0 + np.exp(-x)) * self.args.get("ex_loss_weight1", 0.001)

    # ---- logging ------------------------------------------------------------------------------------------------
    @property
    def iters(self):
        return self._iters

    @property
    def _log_count(self):
        """{"train": triplets of this rank's last training batch, "test": ... validation batch} (weights of the cross-rank means)."""
        return self.__dict__.setdefault("_log_count_d", {})

    def _note_batch(self, batch_item, eval_type):
        t = batch_item.get("slice_between", None) if hasattr(batch_item, "get") else None
        if t is None:
            t = batch_item["image"]
        self._log_count[eval_type] = int(t.shape[0])
        seen = self.__dict__.setdefault("_log_seen", {})
        seen[eval_type] = seen.get(eval_type, 0) + int(t.shape[0])      # triplets behind this rank's epoch means (reset_losses clears it)

    _capture_sink = None      # dict while a step is being captured into a HIP graph (see AEBaseTrainer._train_graphed)

    def _set_mode(self, training):
        """``self.model.train(training)`` without walking ~35 sub-modules on every step when the mode is already set."""
        m = self.model
        if m.training != bool(training) or any(c.training != bool(training) for c in m.children()):
            m.train(bool(training))

    def _log(self, key, value, is_test=False):
        if self._capture_sink is not None and not is_test:
            self._capture_sink[key] = value
            return
        (self.losses_test if is_test else self.losses)[key].append(_scalar(value))

    def reset_losses(self):
        self.__dict__["_log_seen"] = {}
        for d in (self.losses, self.losses_test):
            for k in list(d.keys()):
                d[k] = LossLog()

    def init_tensorboard(self, output_directory):
        self.tb_writer = None
        if self.args.get("log_tensorboard"):
            from torch.utils.tensorboard.writer import SummaryWriter    # optional dependency
            self.tb_writer = SummaryWriter(log_dir=os.path.join(output_directory, "tb"), comment=str(self.args))

    def show_loss_on_tensorboard(self, eval_type="train"):
        if eval_type == "train":
            src, dst = self.losses, self.mean_losses
            self.loss_iters.append(self.iters)
        else:
            src, dst = self.losses_test, self.mean_losses_test
        self._bounded_sync()        # a dead peer ends this rank with a message before the unbounded .item() reads below
        _check_watchdogs(self, "show_loss_on_tensorboard(%s)" % eval_type)      # the means below are what model selection reads
        dp = getattr(self, "dp", None)
        means = {}
        for key, vals in src.items():
            vals = LossLog(vals).floats()
            means[key] = float(np.mean(np.array(vals))) if len(vals) else float("nan")
        if dp is not None and dp.active and means:
            # data parallel: every rank logged the mean over ITS shard; model selection and the loss files must see the mean
            # over the global batch (= what the single-process run logs): sum_r n_r * mean_r / sum_r n_r, one all-reduce
            # weight = the triplets that went into this rank's means since reset_losses() (uneven shards, partial last batches).
            # COLLECTIVE: under data parallel every rank must call this with the same keys (a rank-0-only caller would hang).
            n_local = float(self.__dict__.get("_log_seen", {}).get(eval_type, 0) or self._log_count.get(eval_type, 0) or 0)
            means = dp.reduce_means(means, n_local)
        for key, mean_value in means.items():
            if self.args.get("log_tensorboard") and getattr(self, "tb_writer", None) is not None:
                self.tb_writer.add_scalar("{}/{}".format(key, eval_type), mean_value, self.iters)
            dst[key].append(mean_value)

    def add_image_tensorboard(self, image, log_type):
        if getattr(self, "tb_writer", None) is not None:
            self.tb_writer.add_image(log_type, image, self.iters)

    # ---- losses --------------------------------------------------------------------------------------------------
    def get_loss(self, reference, recons, is_test=False, store_loss=True):
        """Reconstruction loss (reference :164-198): MSE mean, or LPIPS mean with ``--use_percept_loss``."""
        if self.ae_loss_func == "perceptual" and self.percept_criterion is not None:
            if is_test:
                with torch.no_grad():
                    dist = self.percept_criterion(recons, reference, normalize=True).mean()
            else:
                dist = self.percept_criterion(recons, reference, normalize=True).mean()
        else:
            dist = ops.mse_loss(recons, reference)
        lap = 0
        if self.laploss is not None:                                       # reference :183-196
            lap = self.laploss(recons, reference)
        if store_loss:
            self._log("loss_ae_dist", dist, is_test)
            if self.laploss is not None:
                self._log("loss_laploss", lap, is_test)
        return {"loss_ae": dist + lap if self.laploss is not None else dist, "loss_ae_dist": dist, "loss_laploss": lap}

    def _mix_coefficients(self, batch_item, B):
        """(alpha_from, alpha_to) for the latent lerp: 0.5/0.5 here (reference :348-351, trainer_ae.py:51)."""
        return 0.5, 0.5

    def _get_mixup_latent(self, **kwargs):
        z = kwargs.get("z")
        a_from, a_to = kwargs.get("alpha_from", None), kwargs.get("alpha_to", None)
        if a_from is None or not self._per_sample_alpha:
            a_from, a_to = 0.5, 0.5
        return ops.lerp_mix(z, a_from, a_to)

    _per_sample_alpha = False

    def _get_mixup_image(self, **kwargs):
        is_test = kwargs.get("is_test", False)
        z_mix = self._get_mixup_latent(**kwargs)
        if is_test:
            with torch.no_grad():
                s = self.model.decode(z_mix)
        else:
            s = self.model.decode(z_mix)
        return {"z_mix": z_mix, "slice_inbetween_mix": s}

    def get_latent_loss(self, **kwargs):
        reference, no_grad = kwargs.get("reference"), kwargs.get("no_grad", False)
        z_mix = self._get_mixup_latent(**kwargs)
        z_ref = self.encode(reference, eval=True) if no_grad else self.encode(reference, eval=False)
        return {"loss_latent": ops.mse_loss(z_mix.detach(), z_ref.detach()), "z_mix": z_mix}

    # ---- eval wrappers (reference :216-336; device tensors returned, callers .detach().cpu()) ---------------------
    def _use_sr_model(self, use_sr_model=False, **kwargs):
        return self.model_sr if (use_sr_model and self.model_sr is not None) else self.model

    def _to_device(self, t):
        return t.to(self.args["device"]).float()

    def _run_eval(self, model, fn, x, eval, upscale=1, what=None):
        model.train(not eval)
        if eval:
            # eval-mode BatchNorm makes results independent of the batch composition: big batches are cut so that no
            # activation tensor of a pass exceeds the kernels' 32-bit offset range (reference: chunk_size=16 host loop) -- 2^28
            # elements per tensor, against the largest one the compiled stacks really make where the network can say it
            # (HipAE.max_elems_per_image), else against 64 channels at the full output size
            probe = getattr(model, "max_elems_per_image", None)
            per = probe(what, tuple(x.shape[1:])) if (probe is not None and what is not None and x.dim() == 4) else None
            if not per:
                per = x.shape[-2] * upscale * x.shape[-1] * upscale * 64
            n_max = max(1, (1 << 28) // int(per))
            with torch.no_grad():
                if x.shape[0] <= n_max:
                    return fn(x)
                return torch.cat([fn(x[i:i + n_max]) for i in range(0, x.shape[0], n_max)], dim=0)
        return fn(x)

    def predict(self, x, eval=True, chunk_size=16, clear_cache=False, **kwargs):
        return self._run_eval(self.model, self.model, self._to_device(x), eval, what="forward")

    def encode(self, x, eval=True, clear_cache=False, chunk_size=16, **kwargs):
        model = self._use_sr_model(kwargs.get("use_sr_model", False))
        return self._run_eval(model, model.encode, self._to_device(x), eval, what="encode")

    def decode(self, z, eval=True, clear_cache=False, chunk_size=16, **kwargs):
        model = self._use_sr_model(kwargs.get("use_sr_model", False))
        scales = getattr(model, "scales", None)
        if scales is None:
            from ..networks.acai_vanilla import num_scales
            scales = num_scales(self.args)
        return self._run_eval(model, model.decode, self._to_device(z), eval, upscale=1 << int(scales), what="decode")

    # ---- validation (reference :67-99) -----------------------------------------------------------------------------
    def _bounded_sync(self):
        """Data parallel: wait for the queued steps with a deadline (``DataParallelContext.synchronize``, AESR_STEP_TIMEOUT) before any
        unbounded host read (``.item()``, ``.cpu()``, a symbol copy): the library's communicator has no watchdog thread, so a peer that
        died mid-run would otherwise hang this rank inside an RCCL kernel; on timeout the communicator is aborted and the run exits
        non-zero.  Called where the training loop syncs anyway: validation, epoch logging, checkpoints."""
        dp = getattr(self, "dp", None)
        if dp is not None and dp.active:
            dp.synchronize()

    def validate(self, validation_batch, image_dict=None, frame_id=8, generate_images=True):
        self._bounded_sync()
        self.model.eval()
        self._note_batch(validation_batch, "test")
        image = self._to_device(validation_batch["image"])
        z = self.encode(image, eval=True)
        img_recons = self.decode(z, eval=True)
        loss = self.get_loss(image, img_recons, is_test=True)["loss_ae"]
        a_from = validation_batch.get("alpha_from", None)
        a_to = validation_batch.get("alpha_to", None)
        lat = self.get_latent_loss(reference=self._to_device(validation_batch["slice_between"]), z=z, alpha_from=a_from,
                                   alpha_to=a_to, no_grad=True)
        self.test_predictions = {"z": z.detach().cpu(), "img_recons": img_recons.detach().cpu(), "z_device": z.device}
        self._log("loss_ae", loss, True)
        self._log("loss_latent_1", lat["loss_latent"], True)
        if self.epoch > self.args["epoch_threshold"] and "vae" not in self.args.get("model", ""):
            self.save_best_val_model()
        grid = None
        if generate_images:
            from .acai_utils import generate_recon_grid
            grid = generate_recon_grid(validation_batch["image"], self.test_predictions["img_recons"])
        result = {"img_grid_recons": grid, "loss_ae": self.losses_test["loss_ae"][-1]}
        if image_dict is not None:
            vols, alphas = self._generate_val_volumes(image_dict, frame_id=frame_id)
            result.update(synthesized_vols=vols, alphas=alphas)
        _check_watchdogs(self, "validate")      # behind the whole-volume previews too: nothing of this call is handed out unchecked
        return result

    def _generate_val_volumes(self, image_dict, frame_id):
        """Whole-volume validation previews (reference :149-162): for every in-memory 4-D patient ``image_dict[p_id]`` (keys ``image``
        [t,z,y,x], ``patient_id``, ``spacing``) frame ``frame_id`` is cropped to ``eval_patch_size`` (default: ``width``), every 2nd slice
        kept, the held-out slices synthesised at alpha 0.5 and all kept ones reconstructed (``evaluate.evaluate_image.evaluate_image``: one
        encoder pass, one decoder pass per volume); returns ({p_id: comparison grid}, {p_id: alphas}).  A ``frame_id`` beyond the
        patient's last frame means the last frame (``evaluate_image`` clips it, :50-51; the reference then fails on its own dict key)."""
        from collections import defaultdict
        from ..evaluate.evaluate_image import create_compare_image, evaluate_image
        vols, alphas = defaultdict(dict), defaultdict(dict)
        eval_patch_size = self.args.get("eval_patch_size", self.args["width"])
        for p_id, data in image_dict.items():
            f_id = min(int(frame_id), int(data["image"].shape[0]) - 1)
            res = evaluate_image(self, data, frame_id=f_id, downsample_steps=2, eval_patch_size=eval_patch_size)
            vols[p_id] = create_compare_image(res["orig_images"][f_id], res["synth_images"][f_id])
            alphas[p_id] = res["pred_alphas"][f_id]
        return vols, alphas

    def _best_now(self, key):
        hist = self.mean_losses_test[key]
        return len(hist) > 1 and int(np.argmin(hist)) + 1 == len(hist)

    def save_best_val_model(self, **kwargs):
        if self._best_now("loss_ae_dist"):
            self.save_models(os.path.join(self.args["dir_models"], "ae.models"), self.epoch + 1)

    # ---- checkpoints (format: SURVEY App. B) ---------------------------------------------------------------------
    def _is_writer(self):
        return int(os.environ.get("RANK", "0")) == 0

    def save_models(self, fname, epoch):
        self._bounded_sync()
        _check_watchdogs(self, "save_models(%s)" % fname)        # every rank: a checkpoint of garbage is never written
        if not self._is_writer():
            return
        sd = {k: v.detach().cpu().contiguous() for k, v in self.model.state_dict().items()}
        torch.save({"model_dict_ae": sd, "optimizer_dict_ae": optimizer_state_to_cpu(self.opt_ae), "epoch": epoch}, fname)

    def load(self, fname):
        state = torch.load(fname, map_location="cpu")
        self.model.load_state_dict(state["model_dict_ae"])
        if "optimizer_dict_ae" in state and not self.eval_model:
            self.opt_ae.load_state_dict(state["optimizer_dict_ae"])
        print("INFO - {} Loaded model parameters from {}".format(self.__class__.__name__, fname))

    def load_caisr(self, fname):
        state = torch.load(fname, map_location="cpu")
        self.model_sr.load_state_dict(state["model_dict_ae"])

    def save_model(self, **kwargs):
        epoch = kwargs.get("epoch")
        name = "{:0d}_{}.models".format(epoch, self.iters) if kwargs.get("with_iters", False) else "{:0d}.models".format(epoch)
        self.save_models(os.path.join(self.args["dir_models"], name), epoch)

    def save_losses(self):
        if not self._is_writer():
            return
        out = self.args["output_dir"]
        np.savez(os.path.join(out, "loss_iters.npz"), loss_iters=np.array(self.loss_iters))
        np.savez(os.path.join(out, "losses_train.npz"), **{k: np.array(v) for k, v in self.mean_losses.items()})
        np.savez(os.path.join(out, "losses_test.npz"), **{k: np.array(v) for k, v in self.mean_losses_test.items()})

    @staticmethod
    def load_losses(path_to_exper):
        path_to_exper = os.path.expanduser(path_to_exper)
        iters = np.load(os.path.join(path_to_exper, "loss_iters.npz"))["loss_iters"]
        tr = np.load(os.path.join(path_to_exper, "losses_train.npz"))
        te = np.load(os.path.join(path_to_exper, "losses_test.npz"))
        return iters, {k: tr[k] for k in tr.files}, {k: te[k] for k in te.files}

    # ---- epoch hooks ---------------------------------------------------------------------------------------------
    def generate_train_images(self, **kwargs):
        if not self._is_writer() or self.train_predictions is None:
            return
        from .acai_utils import save_image_grid, generate_batch_compare_grid
        grid = generate_batch_compare_grid(kwargs.get("batch_item"), self.train_predictions["slice_inbetween_mix"],
                                           self.train_predictions["reconstruction"])
        save_image_grid(grid, os.path.join(self.args["dir_images"], "train_image_e{:03d}_{}.png".format(kwargs.get("epoch"), self.iters)))

    def end_epoch_processing(self, **kwargs):
        epoch, val = kwargs.get("epoch"), kwargs.get("val_result_dict") or {}
        if self.epoch > self.args["epoch_threshold"]:
            self.save_models(os.path.join(self.args["dir_models"], "{:0d}.models".format(epoch)), epoch)
        self.save_losses()
        if self._is_writer():
            from .acai_utils import save_image_grid
            # example validation volumes, one PNG per patient (reference :416-418)
            for p_id, grid in (val.get("synthesized_vols") or {}).items():
                save_image_grid(grid, os.path.join(self.args["dir_images"], "val_image_e{:03d}_p{:03d}.png".format(epoch, int(p_id))))
            if val.get("img_grid_recons") is not None:
                save_image_grid(val["img_grid_recons"], os.path.join(self.args["dir_images"], "val_recons_e{:03d}.png".format(epoch)))
        dp = getattr(self, "dp", None)
        if dp is not None and dp.active and dp.world > 1:
            # the writer-only work above (checkpoint, npz, PNGs on a possibly slow filesystem) must not run into the next step's in-kernel peer
            # waits (AESR_SYNCBN=p2p gives a late peer ~7 s, not AESR_STEP_TIMEOUT): every rank meets here on the control plane first
            dp.barrier()
        self.epoch += 1      # initialised with 0 in AEBaseTrainer.__init__
