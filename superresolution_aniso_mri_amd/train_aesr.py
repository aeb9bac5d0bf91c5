"""Training loop of ``train_cardiac_aesr.py`` / ``train_brain_aesr.py`` (reference train_cardiac_aesr.py:119-214,
train_brain_aesr.py:137-206) for the MI355X build: same flags, output layout (settings.yaml, models/<epoch>.models,
log_images/, loss*.npz) and epoch bookkeeping (iters starts at 1, validate when (iters+1) % num_it_per_epoch == 0).

Data: the reference's SimpleITK/NIfTI dataset readers and numpy augmentations are host-side I/O outside this build's
hot path; ``--synthetic`` feeds device-resident synthetic triplets in the same batch layout.  Launch under
``python -m torch.distributed.run --nproc-per-node N`` for data parallel training (one process per GPU, RCCL)."""
import os
from shutil import rmtree

import numpy as np
import torch

from .data_synth import shard_batch, synthetic_batch
from .kwatsch.arguments import parse_args
from .kwatsch.common import saveExperimentSettings
from .kwatsch.get_trainer import get_trainer_dynamic
from .networks.net_config import NetworkConfig
from .parallel import DataParallelContext


def merge_args_architecture(args_dict, architecture):
    """CLI wins unless the CLI value is None / absent (reference train_cardiac_aesr.py:23-30)."""
    for key, value in architecture.items():
        if key not in args_dict or args_dict[key] is None:
            args_dict[key] = value
    return args_dict


def prepare_run(args, args_dict, writer):
    out = args_dict["output_dir"]
    if writer:
        if not os.path.isdir(out):
            os.makedirs(out, exist_ok=False)
        elif args_dict["exper_id"] == "debug":
            print("WARNING - prepare_run - Removing output dir {}".format(out))
            rmtree(out)
            os.makedirs(out, exist_ok=False)
        else:
            raise IsADirectoryError("ERROR - directory {} for experiment {} already exists. Remove first or choose another "
                                    "exper_id".format(out, args_dict["exper_id"]))
    args_dict["dir_images"] = os.path.join(out, "log_images")
    args_dict["dir_models"] = os.path.join(out, "models")
    if writer:
        os.makedirs(args_dict["dir_images"], exist_ok=True)
        os.makedirs(args_dict["dir_models"], exist_ok=True)
        saveExperimentSettings(args, os.path.join(out, "settings.yaml"))


def generate_epoch_range(args_dict):
    if args_dict.get("model_filename"):
        last = torch.load(os.path.expanduser(args_dict["model_filename"]), map_location="cpu")["epoch"]
        return np.arange(last + 1, last + args_dict["epochs"] + 1)
    return np.arange(1, args_dict["epochs"] + 1)


def main(argv=None, brain=False):
    args, args_dict = parse_args(argv)
    cfg = NetworkConfig(args_dict["model"], dataset=args_dict["dataset"], ae_class=args_dict["ae_class"])
    args_dict = merge_args_architecture(args_dict, cfg.architecture)
    if not args_dict.get("synthetic") and not args_dict.get("volumes_dir"):
        raise NotImplementedError(
            "the per-dataset readers of the reference (datasets/*) are outside this build's hot path; run with --volumes_dir DIR "
            "(volumes as .npy / .nii / .mha: on-device triplet assembly + augmentation), --synthetic (synthetic triplets in the "
            "same batch layout) or feed trainer.train() from your own loader (dataset '{}')".format(args_dict["dataset"]))
    dp = DataParallelContext()
    if str(args_dict["device"]).startswith("cuda") and dp.world > 1:
        # AESR_SINGLE_DEVICE=1 (with AESR_DIST_BACKEND=gloo): several ranks REHEARSED on one device, as bench.py does
        args_dict["device"] = "cuda:%d" % (0 if os.environ.get("AESR_SINGLE_DEVICE") == "1" else dp.local_rank)
    if str(args_dict["device"]).startswith("cuda"):
        dev = torch.device(args_dict["device"])
        torch.cuda.set_device(dev.index if dev.index is not None else 0)
    dp.device = args_dict["device"]
    torch.manual_seed(args_dict["seed"])
    prepare_run(args, args_dict, writer=(dp.rank == 0))
    brain = brain or args_dict["dataset"] not in ("ACDC", "ACDCC", "ACDCLBL")
    size = args_dict.get("synthetic_size") or args_dict["width"]
    B = args_dict["batch_size"]

    augmenter = None
    if args_dict.get("volumes_dir"):
        # device-resident volume cache + ONE kernel per batch (data_device.py); every rank draws the same global batch from
        # the same RandomState and keeps its own triplets, so the step sees the batch a single process would see
        from .data_device import TripletAugmenter, load_volume_dir
        augmenter = TripletAugmenter(load_volume_dir(args_dict["volumes_dir"]), args_dict["width"],
                                     args_dict.get("aug_patch_size") or args_dict["width"],
                                     rs=np.random.RandomState(args_dict["seed"]), device=args_dict["device"])
        step = max(2, int(args_dict.get("slice_step") or 2))

    synth_pool = {}       # --synthetic: a small pool of device-resident batches made BEFORE the loop (rendering one on the host costs tens of ms)

    def make_batch(seed, n, training=True):
        if augmenter is not None:
            # training batches go into the augmenter's persistent output buffer, which the captured step reads directly (data parallel: the
            # rank's own triplets only -- every rank draws the whole global batch's random numbers); the validation batch is kept for the
            # whole run and gets tensors of its own
            return augmenter.next_batch(n, step=step, reuse_output=training, shard=(dp.rank, dp.world) if dp.active else None)
        elif training:
            key = (n, int(seed) % max(1, int(args_dict.get("synthetic_pool") or 8)))
            if key not in synth_pool:
                b = synthetic_batch(n, size, size, seed=args_dict["seed"] + key[1], brain=brain)
                b = shard_batch(b, dp.rank, dp.world) if dp.active else b
                synth_pool[key] = {k: (v.to(args_dict["device"]) if torch.is_tensor(v) else v) for k, v in b.items()}
            return synth_pool[key]
        else:
            b = synthetic_batch(n, size, size, seed=seed, brain=brain)
        return shard_batch(b, dp.rank, dp.world) if dp.active else b

    trainer = get_trainer_dynamic(args_dict, model_file=args_dict["model_filename"])
    if dp.active:
        dp.attach(trainer)
        dp.set_batch(B)
    # everything built so far lives for the whole run: take it out of the garbage collector's sight, so that a full collection does not
    # walk the module / ctypes object graph in the middle of training (measured: one 95 ms pause around step 100 of a 0.9 ms step)
    import gc
    gc.collect()
    gc.freeze()
    if args_dict.get("use_step_graph"):
        # single process: one HIP graph per step; data parallel: graph segments between the eager collectives
        trainer.enable_step_graph(dp_segments=dp.active)
    trainer.init_tensorboard(args_dict["output_dir"])
    validation_batch = make_batch(args_dict["seed"] - 1, args_dict["test_batch_size"], training=False)
    image_dict_val = None       # train_cardiac_aesr.py:49-53,183: a few in-memory 4-D patients, previewed as whole volumes at every validation
    if args_dict.get("val_volumes_dir"):
        from .data_device import load_image_dict
        image_dict_val = load_image_dict(args_dict["val_volumes_dir"], args_dict.get("val_patients") or 2)
    num_it_per_epoch = args_dict["iters_per_epoch"]
    args.num_it_per_epoch = num_it_per_epoch
    if dp.rank == 0:
        saveExperimentSettings(args, os.path.join(args_dict["output_dir"], "settings.yaml"))
    epoch, val_result, batch_item = 0, None, None
    try:
        for epoch in generate_epoch_range(args_dict):
            trainer.reset_losses()
            for it in range(num_it_per_epoch):
                batch_item = make_batch(args_dict["seed"] + int(epoch) * 100003 + it, B)
                do_validate = (trainer.iters + 1) % num_it_per_epoch == 0
                trainer.train(batch_item, keep_predictions=do_validate)
                if do_validate:
                    val_result = trainer.validate(validation_batch, image_dict=image_dict_val)
                    trainer.show_loss_on_tensorboard()
                    trainer.show_loss_on_tensorboard(eval_type="test")
                    trainer.generate_train_images(epoch=int(epoch), batch_item=batch_item)
                    if dp.rank == 0:
                        print("epoch {:4d} it {:6d} loss_ae {:.6f} val {:.6f}".format(int(epoch), trainer.iters,
                                                                                     trainer.mean_losses["loss_ae"][-1],
                                                                                     trainer.mean_losses_test["loss_ae"][-1]))
                    trainer.reset_losses()
            trainer.end_epoch_processing(epoch=int(epoch), val_result_dict=val_result or {}, batch_item=batch_item)
        trainer.save_models(os.path.join(args_dict["dir_models"], "{:0d}.models".format(int(epoch))), int(epoch))
    except KeyboardInterrupt:
        print("KeyboardInterrupt - Save model and exit")
        trainer.save_models(os.path.join(args_dict["dir_models"], "{:0d}.models".format(int(epoch))), int(epoch))
    finally:
        # the library-owned communicator has no watchdog: leave it with a bounded wait (parallel.DataParallelContext.synchronize) so
        # that a rank whose peer died exits instead of hanging in the teardown
        dp.shutdown()
    return trainer
