#!/usr/bin/env python
"""Headline benchmark: ``ae_combined`` training slices/s on synthetic ACDC-shaped batches (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W [--config c2|c3]

A step = one ae_combined training step (4 network passes, backward, Adam) over one batch of 12 synthetic triplets
(36 slices of 160x160) that is already resident in HBM.  N > 1: launched by torch.distributed.run, one rank per GPU
over RCCL; the 12 triplets are sharded over the ranks (fixed global batch -> strong scaling).
Rank 0 prints ONE JSON line (see DESIGN.md section "Measurement").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

# algorithmic conv FLOPs per training step (SURVEY.md section 8a / BASELINE.md section 2), B=12, 160x160, scales 2
STEP_GFLOP = {"c2": 330.8, "c3": 894.5, "c4": 2223.5, "c5": 1524.4, "c3_scales3": 959.6}
# BASELINE configs[3], configs[4] on ONE rank (secondary numbers; the headline metric is quoted on c2): name -> (dataset, B, H, width, latent_width)
BRAIN = {"c4": ("OASIS", 16, 220, 64, 16), "c5": ("dHCP", 8, 256, 256, 64)}


def stem_folded_gflop(B, H):
    """MFMA flops of the reference's op count that no longer run on the matrix cores: the 32->32 3x3 convolution behind the
    encoder stem is folded with the 1x1 stem into one bandwidth-bound 1->32 convolution (csrc/conv_thin.hip).  That layer ran
    forward on 3B images (x[2B] and the logging-only slice_between[B]) and its data and weight gradients on 2B images each:
    7B image passes x (H+2)^2 x 32*32*9*2 flop = 40.63 GF at B=12, H=160 (330.8 -> 290.17 GF executed)."""
    return 7.0 * B * (H + 2) ** 2 * 32 * 32 * 9 * 2 / 1e9


PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak


def build_args(config, device, scales3=False):
    from superresolution_aniso_mri_amd.networks.net_config import NetworkConfig
    args = dict(model="ae_combined", dataset="ACDC", device=device, lr=1e-5, weight_decay=0.0, epochs=900, batch_size=12,
                width=128, latent_width=32, depth=32, latent=128, ex_loss_weight1=0.05, use_percept_loss=False,
                get_masks=False, use_loss_annealing=False, use_extra_latent_loss=False, epoch_threshold=500,
                ae_class="VanillaACAI", downsample_steps=2, seed=892372,
                image_mix_loss_func="mse" if config == "c2" else "perceptual", vgg_weights="synthetic-hash")
    if config in BRAIN:
        ds, B, _, width, lw = BRAIN[config]
        args.update(dataset=ds, batch_size=B, width=width, latent_width=lw, ex_loss_weight1=0.001)
    if scales3:
        args.update(latent_width=16)       # README-literal latent_width: width // latent_width = 8 -> three pooling stages (C3')
    for k, v in NetworkConfig("ae_combined", dataset=args["dataset"]).architecture.items():
        args.setdefault(k, v)
    return args


def cpu_baseline(B, H, steps=3):
    """The oracle (PyTorch-CPU restatement of the same step, MSE synthesis loss) on this host's cores."""
    from oracle import ae_oracle, step_oracle
    from superresolution_aniso_mri_amd.data_synth import synthetic_batch
    # a 1-GPU box gives this process a 16-CPU share of the host: more intra-op threads than that only oversubscribe
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    torch.set_num_threads(max(1, min(16, avail)))
    torch.manual_seed(892372)
    ae = ae_oracle.OracleAE(ae_oracle.acdc_args())
    st = step_oracle.OracleStep(ae, lr=1e-5, ex_loss_weight1=0.05, image_mix_loss_func="mse")
    batch = synthetic_batch(B, H, H, seed=892372)
    st.train(batch["image"], batch["slice_between"])            # warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        st.train(batch["image"], batch["slice_between"])
    dt = (time.perf_counter() - t0) / steps
    return {"value": round(3 * B / dt, 2), "unit": "slices/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d steps (after 1 warm-up) of the same workload: B=%d triplets %dx%d, MSE synthesis loss, "
                      "oracle/step_oracle.py on PyTorch-CPU fp32" % (steps, B, H, H), "s_per_step": round(dt, 3)}


def make_line(opt, B, H, elapsed, launch, loss, roofline, engine):
    """The JSON line of the contract for ``opt.steps`` steps that took ``elapsed`` seconds (max over ranks)."""
    ms_per_step = 1e3 * elapsed / opt.steps
    executed = STEP_GFLOP[opt.config] - (stem_folded_gflop(B, H) if engine.FUSE_STEM else 0.0)
    line = {
        "metric": "training slices/sec (ae_combined, %dx%d, latent=128)" % (H, H), "value": round(3 * B * opt.steps / elapsed, 1),
        "unit": "slices/s", "n_gpus": opt.gpus, "steps": opt.steps, "warmup": opt.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("ACDC synthetic 12x(3x1x160x160) triplets, ae_combined latent=128 depth=32 scales=2, "
                                + ("MSE synthesis loss (BASELINE configs[1])" if opt.config == "c2"
                                   else "LPIPS-VGG synthesis loss lambda=0.05, synthetic backbone weights (BASELINE configs[2])"))
                   if opt.config not in BRAIN else
                   "%s synthetic %dx(3x1x%dx%d) triplets (global batch), ae_combined latent=128 depth=32, LPIPS-VGG synthesis loss lambda=0.001, "
                   "synthetic backbone weights (BASELINE configs[%d])" % (BRAIN[opt.config][0], B, H, H, 3 if opt.config == "c4" else 4),
                   "global_batch_triplets": B, "slices_per_step": 3 * B, "parallelism": "dp%d" % opt.gpus,
                   "init": "reference Initializer, seed 892372, random weights", "launch": launch},
        "step_algorithmic_gflop": STEP_GFLOP[opt.config],
        "step_executed_mfma_gflop": round(executed, 2),
        "step_tflops": round(executed / ms_per_step, 2),
        "step_frac_of_f32_mfma_peak": round(executed / ms_per_step / PEAK_F32_MFMA_TFLOPS, 4),
        "final_loss": round(float(loss), 6),
    }
    if roofline is not None:
        line["roofline"] = roofline
    return line


def secondary_measurements(device, steps=10, warmup=4):
    """The other single-GPU configurations north_star asks for, measured the same way (inputs resident, captured-graph replay,
    K steps between synchronizes) AFTER the headline measurement: BASELINE configs[2] (C3: + LPIPS), C3 with the README-literal
    three pooling stages, and the 256x256 configuration (configs[4] on one rank).  Reported inside the one JSON line."""
    from superresolution_aniso_mri_amd.data_synth import synthetic_batch
    from superresolution_aniso_mri_amd.kwatsch.get_trainer import get_trainer_dynamic
    out = {}
    for name, config, scales3 in (("c3", "c3", False), ("c3_scales3", "c3", True), ("c5", "c5", False)):
        B, H = (12, 160) if config not in BRAIN else (BRAIN[config][1], BRAIN[config][2])
        torch.manual_seed(892372)
        tr = get_trainer_dynamic(build_args(config, device, scales3=scales3))
        tr.enable_step_graph(eager_steps=2)
        pool = []
        for i in range(2):
            b = synthetic_batch(B, H, H, seed=892372 + i, brain=config in BRAIN)
            pool.append({k: (v.to(device) if torch.is_tensor(v) else v) for k, v in b.items()})
        for i in range(warmup):
            tr.train(pool[i % 2], keep_predictions=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            tr.train(pool[i % 2], keep_predictions=False)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / steps
        gf = STEP_GFLOP[name]
        out[name] = {"workload": "%d triplets of %dx%d, LPIPS-VGG synthesis loss%s" % (B, H, H, ", scales 3 (latent_width 16)" if scales3 else ""),
                     "ms_per_step": round(ms, 3), "slices_per_s": round(3 * B / ms * 1e3, 1), "steps": steps, "warmup": warmup,
                     "step_algorithmic_gflop": gf, "step_algorithmic_tflops": round(gf / ms, 2),
                     "frac_of_f32_mfma_peak": round(gf / ms / PEAK_F32_MFMA_TFLOPS, 4), "final_loss": round(float(tr.losses["loss_ae"][-1]), 6)}
        del tr, pool
        torch.cuda.empty_cache()
    return out


def main():
    # stdout must carry exactly ONE line (the JSON): libraries print banners there (RCCL prints its version block on the first
    # collective), so file descriptor 1 points at stderr for the whole run and the JSON line goes to the saved descriptor
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", choices=["c2", "c3", "c4", "c5"], default="c2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--triplets", type=int, default=12, help="global batch in triplets (12 = the BASELINE workload; other values are for experiments only)")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from the host instead of replaying the captured step")
    ap.add_argument("--dp-graph", choices=("on", "off"), default="on",
                    help="N > 1: 'on' (default) replays the step from the captured graph (RCCL data plane: ONE graph with the collectives "
                    "as nodes; gloo rehearsal: graph segments between eager collectives); 'off' launches every kernel from the host. "
                    "A failure in either form ends the run with a non-zero exit code: no other form is substituted")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary single-GPU measurements (c3, c3 with scales 3, c5)")
    opt = ap.parse_args()

    from superresolution_aniso_mri_amd import engine
    from superresolution_aniso_mri_amd.data_synth import shard_batch, synthetic_batch
    from superresolution_aniso_mri_amd.kwatsch.get_trainer import get_trainer_dynamic
    from superresolution_aniso_mri_amd.parallel import DataParallelContext

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if opt.gpus > 1 and world != opt.gpus:
        raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`" % (opt.gpus, opt.gpus))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("AESR_SINGLE_DEVICE") == "1":       # rehearsal on a one-GPU box: every rank on cuda:0 (with AESR_DIST_BACKEND=gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = "cuda:%d" % local_rank
    dp = DataParallelContext(device=device)
    B, H = opt.triplets, 160
    if opt.config in BRAIN:
        B, H = BRAIN[opt.config][1], BRAIN[opt.config][2]
    torch.manual_seed(892372)
    trainer = get_trainer_dynamic(build_args(opt.config, device))
    if dp.active:
        dp.attach(trainer)
        dp.set_batch(B)
    use_graph = not opt.no_graph and (not dp.active or opt.dp_graph == "on")
    if use_graph:
        trainer.enable_step_graph(eager_steps=2)        # data parallel: the form follows the data plane (parallel.DataParallelContext.graph_mode)
    # a small pool of distinct batches, sharded by triplet and resident in HBM before the timed region
    pool = []
    for i in range(4):
        b = synthetic_batch(B, H, H, seed=892372 + i, brain=opt.config in BRAIN)
        if dp.active:
            b = shard_batch(b, dp.rank, dp.world)
        pool.append({k: (v.to(device) if torch.is_tensor(v) else v) for k, v in b.items()})

    def run(n, first=0):
        for i in range(n):
            trainer.train(pool[(first + i) % len(pool)], keep_predictions=False)

    def measure():
        """W untimed steps, then EXACTLY K steps between barrier + synchronize on both sides; max over ranks."""
        run(opt.warmup)
        dp.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(opt.steps, opt.warmup)
        torch.cuda.synchronize()
        dp.barrier()
        return dp.max_over_ranks(time.perf_counter() - t0)

    elapsed = measure()
    launch = "host launches"
    if use_graph:
        launch = ("captured HIP graph replay" if not dp.active else
                  "one HIP graph per step, RCCL collectives (library-owned communicator) as graph nodes" if dp.graph_mode == "whole" else
                  "HIP graph segments between eager host-staged (gloo) collectives")
        if not getattr(trainer, "_graphs", None):
            raise SystemExit("bench: the step graph was requested but never captured")
    loss = trainer.losses["loss_ae"][-1]

    roofline = None
    if not opt.no_roofline:
        # same steps again with one HIP-event pair around every MFMA convolution launch (on the launching stream)
        trainer._graph_enabled = False          # event pairs need host-side launches
        engine.PROFILER = engine.KernelProfiler()
        run(min(opt.steps, 5), 0)
        summ = engine.PROFILER.summary()
        engine.PROFILER = None
        # dominant kernel = the MFMA convolution kernel with the most time in the step.  Kinds are the library's kernels:
        # conv_wino_f32 / conv_wino_ring_f32 / conv_wino_res_f32 (Winograd forward + data gradient: streamed, ring, resident filter), conv_wgrad_wino_f32
        # (Winograd weight gradient), conv_igemm_f32 / conv_wgrad_f32 (direct forms, where the Winograd ones do not apply)
        nst = max(1, min(opt.steps, 5))
        kinds = ("conv_wino_f32", "conv_wino_ring_f32", "conv_wino_res_f32", "conv_wgrad_wino_f32", "conv_igemm_f32", "conv_wgrad_f32")
        kname = max(kinds, key=lambda n: summ.get(n, {"ms": 0.0})["ms"])
        k = summ.get(kname, {"launches": 0, "flops": 0.0, "ms": 1e-9})
        ach = k["flops"] / (k["ms"] * 1e-3) / 1e12 if k["launches"] else 0.0
        wino = "wino" in kname
        roofline = {"bound": "mfma", "kernel": kname, "achieved": round(ach, 2),
                    "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                    "traffic": None, "launches_per_step": k["launches"] // nst,
                    "avg_launch_us": round(1e3 * k["ms"] / max(1, k["launches"]), 2),
                    "algorithmic_gflop_per_launch": round(k["flops"] / max(1, k["launches"]) / 1e9, 3),
                    "kernel_ms_per_step": round(k["ms"] / nst, 3),
                    "dtype": "f32 (v_mfma_f32_16x16x4_f32)",
                    "flop_basis": ("algorithmic (direct 3x3 convolution: 2*N*H*W*Cout*9*Cin per launch); the kernel is Winograd F(2x2,3x3) "
                                   "and executes 1/2.25 of these on the matrix cores, so the fraction of the MFMA peak can exceed 1; "
                                   "executed_frac = frac / 2.25" if wino else "algorithmic = executed"),
                    "measured": "HIP events around every launch, %d instrumented steps after the timed region" % nst}
        if wino:
            roofline["executed_frac"] = round(ach / 2.25 / PEAK_F32_MFMA_TFLOPS, 4)
        roofline["other_kernels"] = [
            {"kernel": n, "achieved": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2), "ms_per_step": round(v["ms"] / nst, 3),
             "launches_per_step": v["launches"] // nst}
            for n, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"]) if n != kname and n in kinds and v["launches"]]
        # HBM-side bytes per launch of the same kernel: PMC counters cannot be read from inside this process; they come from the
        # committed rocprofv3 --pmc passes of this very command (scripts/pmc_traffic.py -> profiles/), N=1 and B=12 only
        tpath = os.path.join(ROOT, "profiles", "r02_%s_hbm_traffic.json" % opt.config)
        if opt.gpus == 1 and B == 12 and opt.config in ("c2", "c3") and os.path.exists(tpath):
            tk = json.load(open(tpath))["kernels"]
            ig = [v for name, v in tk.items() if (kname + "<") in name or (kname + "(") in name]
            nl = sum(v["launches_per_step"] for v in ig)
            if nl > 0:
                roofline["traffic"] = round(sum(v["MB_per_step"] for v in ig) / nl * 1e6)
                roofline["traffic_unit"] = "bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/r02_%s_hbm_traffic.json)" % opt.config

    if dp.rank != 0:
        dp.shutdown()
        return
    line = make_line(opt, B, H, elapsed, launch, loss, roofline, engine)
    if opt.gpus == 1 and not opt.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(B, H)
    sys.stdout.flush()
    if opt.gpus == 1 and opt.config == "c2" and B == 12 and not opt.no_secondary:
        del trainer, pool
        torch.cuda.empty_cache()
        line["secondary"] = secondary_measurements(device)
    os.write(real_stdout, (json.dumps(line) + "\n").encode())
    dp.shutdown()


if __name__ == "__main__":
    main()
