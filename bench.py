#!/usr/bin/env python
"""Headline benchmark: ``ae_combined`` training slices/s on synthetic ACDC-shaped batches (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W [--config c2|c3|c4|c5]

A step = one ae_combined training step (4 network passes, backward, Adam) over one batch of 12 synthetic triplets
(36 slices of 160x160) that is already resident in HBM.  N > 1: launched by torch.distributed.run, one rank per GPU
over RCCL; the 12 triplets are sharded over the ranks (fixed global batch -> strong scaling).
Rank 0 prints ONE JSON line (see DESIGN.md section "Measurement").
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def self_launch(argv):
    """``python bench.py --gpus N`` (N > 1) without an outer launcher: start the N ranks as FRESH child processes under
    ``torch.distributed.run`` -- before this process imports torch or touches a GPU, never by exec -- let rank 0's JSON line through on the
    inherited stdout and return the launcher's exit code.  Returns None when there is nothing to launch (N = 1, or the ranks' environment
    is already there: the driver's own ``python -m torch.distributed.run ... bench.py`` form)."""
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument("--gpus", type=int, default=1)
    gpus = ap.parse_known_args(argv)[0].gpus
    if gpus <= 1 or "WORLD_SIZE" in os.environ:
        return None
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:        # a free rendezvous port on the loop-back interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # the host driver only supports dmabuf IPC (RCCL across processes needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    sys.stderr.write("bench: launching %d ranks: %s\n" % (gpus, " ".join(cmd)))
    sys.stderr.flush()
    return subprocess.run(cmd, env=env).returncode


if __name__ == "__main__":
    _rc = self_launch(sys.argv[1:])
    if _rc is not None:
        sys.exit(_rc)

import torch  # noqa: E402

# algorithmic conv FLOPs per training step (SURVEY.md section 8a / BASELINE.md section 2)
STEP_GFLOP = {"c2": 330.8, "c3": 894.5, "c4": 2223.5, "c5": 1524.4, "c3_scales3": 959.6}
# BASELINE configs[3], configs[4] on ONE rank (secondary numbers; the headline metric is quoted on c2): name -> (dataset, B, H, width, latent_width)
BRAIN = {"c4": ("OASIS", 16, 220, 64, 16), "c5": ("dHCP", 8, 256, 256, 64)}
WORKLOAD = {
    "c2": "ACDC synthetic 12x(3x1x160x160) triplets, ae_combined latent=128 depth=32 scales=2, MSE synthesis loss (BASELINE configs[1])",
    "c3": "ACDC synthetic 12x(3x1x160x160) triplets, ae_combined latent=128 depth=32 scales=2, LPIPS-VGG synthesis loss lambda=0.05, "
          "synthetic backbone weights (BASELINE configs[2])",
    "c4": "OASIS synthetic 16x(3x1x220x220) triplets (global batch), ae_combined latent=128 depth=32, LPIPS-VGG synthesis loss lambda=0.001, "
          "synthetic backbone weights (BASELINE configs[3] on one rank)",
    "c5": "dHCP synthetic 8x(3x1x256x256) triplets (global batch), ae_combined latent=128 depth=32, LPIPS-VGG synthesis loss lambda=0.001, "
          "synthetic backbone weights (BASELINE configs[4] on one rank)",
}
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
PEAK_HBM_TBPS = 8.0               # MI355X_MICROARCH.md: HBM3E
# tail kinds of engine.KernelProfiler -> prefixes of the rocprofv3 kernel names that serve them (profiles/rNN_<cfg>_kernel_stats.json)
TAIL_ROCPROF = {
    "bn_fwd": ("bn_stats_kernel", "bn_reduce_finalize_kernel", "bn_apply_kernel", "bn_fused_fwd_kernel", "bn_finalize_kernel", "bn_reduce_kernel"),
    "bn_bwd": ("bn_bwd_reduce_kernel", "bn_bwd_reduce_finalize_kernel", "bn_bwd_apply_kernel", "bn_bwd_kernel", "bn_fused_bwd_kernel"),
    "thin_expand": ("thin_expand_kernel",), "thin_collapse": ("thin_collapse_kernel",),
    "thin_reduce": ("thin_reduce_kernel", "thin_stem_finish_kernel", "thin_cout1_finish_kernel"),
    "wgrad_reduce_many": ("wgrad_reduce_many_kernel",), "mse3": ("mse3_fwd_kernel", "mse3_bwd_kernel"), "lerp_cat": ("lerp_cat_fwd_kernel", "lerp_cat_bwd_kernel"),
    "adam_step": ("adam_step_kernel",), "prep_many": ("prep_many_kernel",), "maxpool2": ("maxpool2_fwd_kernel", "maxpool2_bwd_kernel"),
    "lpips_tap": ("lpips_tap_fwd", "lpips_tap_bwd"), "act_bwd": ("act_bwd_kernel",), "mse": ("sqdiff_partial_kernel",),
}
MFMA_KINDS = ("conv_wino_f32", "conv_wino_ring_f32", "conv_wino_res_f32", "conv_wgrad_wino_f32", "conv_igemm_f32", "conv_wgrad_f32")


def stem_folded_gflop(B, H):
    """MFMA flops of the reference's op count that no longer run on the matrix cores: the 32->32 3x3 convolution behind the
    encoder stem is folded with the 1x1 stem into one bandwidth-bound 1->32 convolution (csrc/conv_thin.hip).  That layer ran
    forward on 3B images (x[2B] and the logging-only slice_between[B]) and its data and weight gradients on 2B images each:
    7B image passes x (H+2)^2 x 32*32*9*2 flop = 40.63 GF at B=12, H=160 (330.8 -> 290.17 GF on the matrix cores)."""
    return 7.0 * B * (H + 2) ** 2 * 32 * 32 * 9 * 2 / 1e9


def csrc_sha():
    """Hash of the kernel sources' CODE (comments and white space removed, so that editing a comment does not orphan a measurement):
    the PMC traffic files under profiles/ carry the hash they were measured at, and a stale one is not quoted (``traffic: null``)."""
    import re
    h = hashlib.sha256()
    d = os.path.join(ROOT, "superresolution_aniso_mri_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            src = open(os.path.join(d, name), "r", errors="replace").read()
            src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)            # block comments
            src = re.sub(r"//[^\n]*", " ", src)                          # line comments (no "//" occurs inside a string of these files)
            h.update(name.encode())
            h.update(" ".join(src.split()).encode())
    return h.hexdigest()[:16]


def rocprof_stats(config, gpus, B):
    """The newest committed rocprofv3 kernel statistics of the replayed step of ``config`` (profiles/rNN_<cfg>_kernel_stats.json, written by
    scripts/round_profiles.sh) whose stamped hash equals the kernel sources this run is on; None otherwise (other sources, N > 1, another batch)."""
    import glob
    if gpus != 1 or B != config_shape(config)[0]:
        return None
    sha = csrc_sha()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_%s_kernel_stats.json" % config)), reverse=True):
        try:
            j = json.load(open(path))
        except (OSError, ValueError):
            continue
        if j.get("csrc_sha") == sha:
            j["file"] = "profiles/" + os.path.basename(path)
            return j
    return None


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


def config_shape(config, triplets=0):
    B, H = (12, 160) if config not in BRAIN else (BRAIN[config][1], BRAIN[config][2])
    return (int(triplets) if triplets else B), H


def cpu_baseline(config, B, H, steps=3):
    """The oracle (PyTorch-CPU restatement of the same step, with the configuration's own synthesis loss) on this host's cores."""
    import numpy as np
    from oracle import ae_oracle, lpips_oracle, step_oracle
    from superresolution_aniso_mri_amd.data_synth import synthetic_batch
    # a 1-GPU box gives this process a 16-CPU share of the host: more intra-op threads than that only oversubscribe
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    torch.set_num_threads(max(1, min(16, avail)))
    torch.manual_seed(892372)
    ae = ae_oracle.OracleAE(ae_oracle.acdc_args())
    kw, lossname = {}, "MSE synthesis loss"
    if config != "c2":
        lin = np.load(os.path.join(ROOT, "superresolution_aniso_mri_amd", "lpips", "weights", "v0.1", "vgg_lin.npz"))
        kw = dict(vgg_sd=lpips_oracle.hash_vgg16_state(), lin_w=[torch.from_numpy(lin["lin%d" % k]).reshape(1, -1, 1, 1) for k in range(5)])
        lossname, steps = "LPIPS-VGG synthesis loss (synthetic backbone)", 2
    st = step_oracle.OracleStep(ae, lr=1e-5, ex_loss_weight1=0.05, image_mix_loss_func="mse" if config == "c2" else "perceptual", **kw)
    batch = synthetic_batch(B, H, H, seed=892372)
    st.train(batch["image"], batch["slice_between"])            # warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        st.train(batch["image"], batch["slice_between"])
    dt = (time.perf_counter() - t0) / steps
    return {"value": round(3 * B / dt, 2), "unit": "slices/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d steps (after 1 warm-up) of the same workload: B=%d triplets %dx%d, %s, oracle/step_oracle.py on PyTorch-CPU fp32"
                      % (steps, B, H, H, lossname), "s_per_step": round(dt, 3)}


def timed_steps(trainer, pool, steps, warmup, dp=None):
    """W untimed steps, then EXACTLY K steps between barrier + synchronize on both sides; max over ranks.  Seconds."""
    def run(n, first=0):
        for i in range(n):
            trainer.train(pool[(first + i) % len(pool)], keep_predictions=False)
    def sync():
        if dp is not None and dp.active:
            dp.synchronize()            # torch.cuda.synchronize() with a deadline: a dead peer ends the run (non-zero exit), no hang
        else:
            torch.cuda.synchronize()
    run(warmup)
    if dp is not None:
        dp.barrier()
    sync()
    t0 = time.perf_counter()
    run(steps, warmup)
    sync()
    if dp is not None:
        dp.barrier()
    dt = time.perf_counter() - t0
    return dp.max_over_ranks(dt) if dp is not None else dt


def make_trainer(config, device, B, H, scales3=False, graph=True, npool=4, dp=None):
    from superresolution_aniso_mri_amd.data_synth import shard_batch, synthetic_batch
    from superresolution_aniso_mri_amd.kwatsch.get_trainer import get_trainer_dynamic
    torch.manual_seed(892372)
    trainer = get_trainer_dynamic(build_args(config, device, scales3=scales3))
    if dp is not None and dp.active:
        dp.attach(trainer)
        dp.set_batch(B)
    if graph:
        trainer.enable_step_graph(eager_steps=2)        # data parallel: the form follows the data plane (parallel.DataParallelContext.graph_mode)
    pool = []       # a small pool of distinct batches, sharded by triplet and resident in HBM before the timed region
    for i in range(npool):
        b = synthetic_batch(B, H, H, seed=892372 + i, brain=config in BRAIN)
        if dp is not None and dp.active:
            b = shard_batch(b, dp.rank, dp.world)
        pool.append({k: (v.to(device) if torch.is_tensor(v) else v) for k, v in b.items()})
    import gc
    gc.collect()
    gc.freeze()         # as train_aesr.py does: no full collection over the long-lived object graph inside the timed steps
    return trainer, pool


def roofline_of(engine, device, nst, config, B, H, gpus, scales3=False):
    """The same steps again, host-launched on a trainer of their own (event pairs need host-side launches; a trainer that has replayed
    graphs keeps its pool and times the first eager launches wrongly), with one HIP-event pair around every MFMA convolution launch
    (on the launching stream).  Dominant kernel
    = the MFMA convolution kernel with the most time in the step.  Kinds are the library's kernels: conv_wino_f32 / conv_wino_ring_f32
    / conv_wino_res_f32 (Winograd forward + data gradient: first streamed kernel, filter ring, resident filter), conv_wgrad_wino_f32
    (Winograd weight gradient), conv_igemm_f32 / conv_wgrad_f32 (direct forms, where the Winograd ones do not apply)."""
    trainer, pool = make_trainer(config, device, B, H, scales3=scales3, graph=False, npool=2)
    for i in range(3):
        trainer.train(pool[i % len(pool)], keep_predictions=False)
    torch.cuda.synchronize()
    engine.PROFILER = engine.KernelProfiler()
    for i in range(nst):
        trainer.train(pool[i % len(pool)], keep_predictions=False)
    summ = engine.PROFILER.summary()
    engine.PROFILER = None
    del trainer, pool
    # what a HIP-event pair reads around NOTHING on this stream (median of 200 empty pairs, ~4.6-5.3 us on MI355X / ROCm 7): every duration above
    # carries it -- 8-9 % of a 55 us launch.  ``achieved`` / ``frac`` stay on the RAW durations (a lower bound on the rate; the basis of rounds
    # 1-4); the figures net of it and the rocprofv3 trace of the same kernel sources stand beside them (round-5 verdict, weak 4 / advice)
    pairs = []
    for _ in range(200):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream())
        e1.record(torch.cuda.current_stream())
        pairs.append((e0, e1))
    torch.cuda.synchronize()
    ov_ms = sorted(a.elapsed_time(b) for a, b in pairs)[len(pairs) // 2]
    kname = max(MFMA_KINDS, key=lambda n: summ.get(n, {"ms": 0.0})["ms"])
    wino = "wino" in kname
    div = 2.25 if wino else 1.0
    prof = rocprof_stats(config, gpus, B)                   # the committed rocprofv3 kernel trace of the same kernel sources, or None

    def mfma_entry(name, ev, full):
        """Rates of one MFMA kernel family: RAW event durations (the basis of achieved / frac, as in rounds 1-4), the same net of the empty
        event pair, and the rocprofv3 trace of the same kernel sources where one is committed -- the arbiter between the two."""
        d = 2.25 if "wino" in name else 1.0
        rate = lambda ms: ev["flops"] / (ms * 1e-3) / 1e12 / d if ms > 0 else 0.0
        n = ev["launches"]
        net_ms = max(ev["ms"] - n * ov_ms, 1e-9)
        e = {"kernel": name, "achieved": round(rate(ev["ms"]), 2), "frac": round(rate(ev["ms"]) / PEAK_F32_MFMA_TFLOPS, 4),
             "avg_launch_us": round(1e3 * ev["ms"] / n, 2), "launches_per_step": n // nst, "kernel_ms_per_step": round(ev["ms"] / nst, 3),
             "algorithmic_achieved": round(rate(ev["ms"]) * d, 2), "algorithmic_frac": round(rate(ev["ms"]) * d / PEAK_F32_MFMA_TFLOPS, 4),
             "net_of_event_overhead": {"avg_launch_us": round(1e3 * net_ms / n, 2), "achieved": round(rate(net_ms), 2),
                                       "frac": round(rate(net_ms) / PEAK_F32_MFMA_TFLOPS, 4)}}
        if prof is not None:
            rows = [v for k_, v in prof["kernels"].items() if (name + "<") in k_ or (name + "(") in k_]
            nl, us = sum(v["n_per_step"] for v in rows), sum(v["us_per_step"] for v in rows)
            if nl > 0 and abs(nl - n / nst) < 0.5:          # the same launches per step: the same work
                r_ms = us * 1e-3 * nst
                e["rocprof"] = {"avg_launch_us": round(us / nl, 2), "achieved": round(rate(r_ms), 2), "frac": round(rate(r_ms) / PEAK_F32_MFMA_TFLOPS, 4)}
        if not full:
            for k_ in ("algorithmic_achieved", "algorithmic_frac"):
                e.pop(k_)
        return e

    k = summ.get(kname, {"launches": 0, "flops": 0.0, "ms": 1e-9})
    roofline = {"bound": "mfma", "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "traffic": None}
    if k["launches"]:
        roofline.update(mfma_entry(kname, k, True))
        roofline["algorithmic_gflop_per_launch"] = round(k["flops"] / k["launches"] / 1e9, 3)
    roofline.update({
        "dtype": "f32 (v_mfma_f32_16x16x4_f32)",
        "flop_basis": ("achieved / frac = flops EXECUTED on the matrix cores: the kernel is Winograd F(2x2,3x3) and runs 1/2.25 of the "
                       "algorithmic flops (direct 3x3 convolution, 2*N*H*W*Cout*9*Cin per launch, SURVEY 8d), on which "
                       "algorithmic_achieved / algorithmic_frac are quoted" if wino else "algorithmic = executed"),
        "measured": ("achieved / frac / avg_launch_us: RAW durations of one HIP-event pair per launch on the launching stream, %d instrumented host-launched "
                     "steps after the timed region (an upper bound on the launch time: an EMPTY pair reads %.2f us on this stream, median of 200); "
                     "net_of_event_overhead: the same minus that reading per launch; rocprof: the kernel's average in the committed rocprofv3 "
                     "--kernel-trace of the replayed step at the same kernel sources (%s) -- the arbiter" % (
                         nst, 1e3 * ov_ms, prof["file"] if prof is not None else "none committed for these sources")),
        "event_pair_overhead_us": round(1e3 * ov_ms, 2)})
    roofline["other_kernels"] = [mfma_entry(n, v, False) for n, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])
                                 if n != kname and n in MFMA_KINDS and v["launches"]]
    # ---- the bandwidth-bound tail of the step: algorithmic bytes (every tensor of a call read / written once) over time, against 8 TB/s ----
    tail, t_bytes, t_us = [], 0.0, 0.0
    for n, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"]):
        if n in MFMA_KINDS or not v["launches"]:
            continue
        net_us = max(v["ms"] - v["launches"] * ov_ms, 1e-6) * 1e3 / nst
        row = {"kind": n, "calls_per_step": round(v["launches"] / nst, 1), "algorithmic_MB_per_step": round(v["bytes"] / nst / 1e6, 2),
               "us_per_step_events": round(1e3 * v["ms"] / nst, 1), "us_per_step_net": round(net_us, 1)}
        us = net_us
        if prof is not None and n in TAIL_ROCPROF:
            rows = [x for k_, x in prof["kernels"].items() if any(k_.startswith(p_) or (" " + p_) in k_ for p_ in TAIL_ROCPROF[n])]
            if rows:
                us = sum(x["us_per_step"] for x in rows)
                row["us_per_step_rocprof"] = round(us, 1)
                row["launches_per_step_rocprof"] = round(sum(x["n_per_step"] for x in rows), 1)
        if v["bytes"] > 0:
            row["GB_per_s"] = round(v["bytes"] / nst / us / 1e3, 1)
            row["frac_of_hbm_peak"] = round(v["bytes"] / nst / us / 1e3 / (PEAK_HBM_TBPS * 1e3), 4)
        row["time_basis"] = "rocprof" if "us_per_step_rocprof" in row else "events net of the empty pair"
        tail.append(row)
        t_bytes += v["bytes"] / nst
        t_us += us
    roofline["tail"] = {"bound": "hbm", "peak": PEAK_HBM_TBPS, "unit": "TB/s", "kinds": tail, "us_per_step": round(t_us, 1),
                        "algorithmic_MB_per_step": round(t_bytes / 1e6, 1),
                        "frac_of_hbm_peak": round(t_bytes / max(t_us, 1e-9) / 1e6 / PEAK_HBM_TBPS, 4) if t_us > 0 else None,
                        "reading": ("every non-MFMA call of the step the library times (BatchNorm forward / backward incl. the fused pooling, the "
                                    "single-channel-side 'thin' convolutions, slab sums, losses, lerp, Adam, weight preparation, the LPIPS head): "
                                    "algorithmic bytes = each tensor of the call once; a call may be several launches (BatchNorm: 2-3), the time is the "
                                    "call's.  torch glue (a few copy / cat kernels, ~20 us per step) is outside")}
    if prof is not None:
        roofline["rocprof_file"] = prof["file"]
        roofline["rocprof_kernel_us_per_step"] = prof.get("kernel_us_per_step")
    # HBM-side bytes per launch of the same kernel: PMC counters cannot be read from inside this process; they come from the
    # committed rocprofv3 --pmc passes of this very command (scripts/round_profiles.sh -> profiles/rNN_<cfg>_hbm_traffic.json), N=1 and B=12 only, and only
    # while the kernel sources are the ones the passes were measured on
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_%s_hbm_traffic.json" % config)), reverse=True)      # newest round first
    if gpus == 1 and B == 12 and cands:
        sha = csrc_sha()
        for tpath in cands:
            tj = json.load(open(tpath))
            if tj.get("csrc_sha") != sha:
                continue
            ig = [v for name, v in tj["kernels"].items() if (kname + "<") in name or (kname + "(") in name]
            nl = sum(v["launches_per_step"] for v in ig)
            if nl > 0:
                roofline["traffic"] = round(sum(v["MB_per_step"] for v in ig) / nl * 1e6)
                roofline["traffic_unit"] = ("bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/%s, kernel sources %s)"
                                            % (os.path.basename(tpath), tj["csrc_sha"]))
            # the tail's kinds: HBM-side MB per step of the kernels that serve them, beside their algorithmic MB (well above = re-reads: the
            # three-launch BatchNorm reads its layer twice)
            for row in roofline["tail"]["kinds"]:
                pre = TAIL_ROCPROF.get(row["kind"], ())
                hit = [v for name, v in tj["kernels"].items() if any(name.startswith(p_) or (" " + p_) in name for p_ in pre)]
                if hit:
                    row["traffic_MB_per_step"] = round(sum(v["MB_per_step"] for v in hit), 1)
            roofline["tail"]["traffic_MB_per_step"] = round(sum(r_.get("traffic_MB_per_step", 0.0) for r_ in roofline["tail"]["kinds"]), 1)
            break
        else:
            roofline["traffic_note"] = "%s was measured on other kernel sources (%s, now %s): not quoted" % (
                os.path.basename(cands[0]), json.load(open(cands[0])).get("csrc_sha"), sha)
    return roofline


def step_numbers(config, B, H, ms_per_step, fuse_stem, scales3=False):
    """Step-level flop rates: the reference's op count (what the 70 % target of north_star is read against) and, beside it,
    what the matrix cores execute (the op count minus the folded stem layer, in Winograd form 1/2.25 of it; the only MFMA layer that is
    not Winograd, VGG conv1_1, is 0.3 % of the C3 step)."""
    gf = STEP_GFLOP[config + "_scales3" if scales3 else config] * (B / float(config_shape(config)[0]))
    executed = (gf - (stem_folded_gflop(B, H) if fuse_stem else 0.0)) / 2.25
    return {"step_algorithmic_gflop": round(gf, 1), "step_algorithmic_tflops": round(gf / ms_per_step, 2),
            "step_algorithmic_frac_of_f32_mfma_peak": round(gf / ms_per_step / PEAK_F32_MFMA_TFLOPS, 4),
            "step_executed_mfma_gflop": round(executed, 2),
            "step_executed_frac_of_f32_mfma_peak": round(executed / ms_per_step / PEAK_F32_MFMA_TFLOPS, 4)}


def secondary_config(engine, config, device, scales3=False, steps=10, warmup=4, with_roofline=False, with_cpu=False):
    B, H = config_shape(config)
    trainer, pool = make_trainer(config, device, B, H, scales3=scales3, npool=2)
    ms = 1e3 * timed_steps(trainer, pool, steps, warmup) / steps
    out = {"workload": WORKLOAD[config] + (", scales 3 (latent_width 16)" if scales3 else ""), "ms_per_step": round(ms, 3),
           "slices_per_s": round(3 * B / ms * 1e3, 1), "steps": steps, "warmup": warmup,
           "final_loss": round(float(trainer.losses["loss_ae"][-1]), 6)}
    out.update(step_numbers(config, B, H, ms, engine.FUSE_STEM, scales3))
    del trainer, pool
    torch.cuda.empty_cache()
    if with_roofline:
        out["roofline"] = roofline_of(engine, device, 3, config, B, H, 1, scales3)
        torch.cuda.empty_cache()
    if with_cpu:
        out["cpu_baseline"] = cpu_baseline(config, B, H)
    return out


# what ONE rank of the 8-rank run of each BASELINE configuration executes: config -> (H, single-process shard sizes, rank-step shard sizes,
# slices of the global batch, triplets on the slowest of 8 ranks).  c2 / c3: 12 triplets -> 2,2,2,2,1,1,1,1; c4 (configs[3]): 16 -> 2 each;
# c5 (configs[4]): 8 -> 1 each (SyncBN is what keeps its statistics those of the global batch)
SHARD_CASES = {"c2": (160, (1, 2, 3, 6), (1, 2), 36, 2), "c3": (160, (1, 2, 3, 6), (1, 2), 36, 2),
               "c4": (220, (2,), (2,), 48, 2), "c5": (256, (1,), (1,), 24, 1)}


def small_shards(device, configs=("c2", "c3", "c4", "c5")):
    """What strong scaling runs on, measured on ONE GPU: the replayed single-process step at the shard sizes of SHARD_CASES and -- what a
    RANK of a data-parallel run executes -- the same step with the data-parallel machinery switched on over a communicator of one
    (``*_dp_ms``: SyncBN partial-sum kernels, the 8 SyncBN + 1 gradient collectives as RCCL enqueues, the default graph form; no wire
    time, no waiting for peers).  The 8-rank rate is bounded by the global batch per step of the slowest rank."""
    import torch.distributed as dist
    from superresolution_aniso_mri_amd.parallel import DataParallelContext
    out = {}
    for config in configs:
        H, singles, _, _, _ = SHARD_CASES[config]
        row = {}
        for t in singles:
            trainer, pool = make_trainer(config, device, t, H, npool=2)
            row["%d_triplets_ms" % t] = round(1e3 * timed_steps(trainer, pool, 20, 6) / 20, 3)
            del trainer, pool
        torch.cuda.empty_cache()
        out[config] = row

    def rank_steps(suffix):
        if not dist.is_initialized():          # a group of one without a network port
            dist.init_process_group("gloo", store=dist.HashStore(), rank=0, world_size=1)
        dp = DataParallelContext(device=device)
        try:
            for config in configs:
                H, _, ranks, _, _ = SHARD_CASES[config]
                for t in ranks:
                    trainer, pool = make_trainer(config, device, t, H, npool=2, dp=dp)
                    out[config]["%d_triplets_%s_ms" % (t, suffix)] = round(1e3 * timed_steps(trainer, pool, 20, 6, dp) / 20, 3)
                    del trainer, pool
                torch.cuda.empty_cache()
            return "communicator of one, data plane %s, graph form %s, SyncBN exchange %s" % (dp.data_backend, dp.graph_mode, dp.syncbn)
        finally:
            dp.shutdown()

    saved = {k: os.environ.get(k) for k in ("AESR_FORCE_DP", "AESR_SYNCBN")}
    try:
        os.environ["AESR_FORCE_DP"] = "1"
        out["dp_form"] = rank_steps("dp")
        # the opt-in one-shot SyncBN exchange (AESR_SYNCBN=p2p: inside the one-launch BatchNorm kernels, over a peer group of one): only the
        # gradient all-reduce is left as a collective
        os.environ["AESR_SYNCBN"] = "p2p"
        out["dp_p2p_form"] = rank_steps("dp_p2p") + " (opt-in)"
    except Exception as e:       # no RCCL on this box: the projection falls back to the single-process step
        out["dp_note"] = "one-rank data-parallel step not measured (%s)" % (str(e)[:160],)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    for config in configs:
        _, _, _, slices, t = SHARD_CASES[config]
        step_ms = out[config].get("%d_triplets_dp_ms" % t, out[config]["%d_triplets_ms" % t])
        out[config]["projected_8_rank_slices_per_s"] = round(slices / step_ms * 1e3, 1)
        if "%d_triplets_dp_p2p_ms" % t in out[config]:
            out[config]["projected_8_rank_slices_per_s_p2p"] = round(slices / out[config]["%d_triplets_dp_p2p_ms" % t] * 1e3, 1)
    out["projection"] = ("global batch (36 / 36 / 48 / 24 slices) / the step of the slowest of 8 ranks (2 / 2 / 2 / 1 triplets) with SyncBN and the 9 "
                         "collectives enqueued on a communicator of one: an UPPER bound on the 8-rank rate (wire time and waiting for the slowest "
                         "peer come on top)")
    return out


def e2e_bench(device, steps=300, warmup=20):
    """The training LOOP, not only the step (train_cardiac_aesr.py:153-204 + datasets/shared_transforms.py:48-120 of the
    reference): as train_aesr.py runs it with --volumes_dir -- per step the host draws the batch's random numbers from a numpy RandomState
    in the reference's order (slice pair, from/to order, crop origin, intensity curve, rotation: ~10 draws per triplet), ONE launch
    (aesr_triplet_assemble) gathers, crops, intensity-maps and rotates the 12 triplets out of a device-resident volume cache into the
    buffer the captured step reads, and the step graph is replayed.  Synthetic ACDC-shaped cache: 60 volumes of 10 x 216 x 256,
    aug_patch_size 180, width 160 (configs[1]).  Beside it, the same trainer fed ONE resident batch (the headline's regime)."""
    import numpy as np
    from superresolution_aniso_mri_amd.data_device import TripletAugmenter
    rs = np.random.RandomState(892372)
    vols = []
    for _ in range(60):
        low = rs.rand(10, 27, 32).astype(np.float32)
        vols.append(np.clip(np.kron(low, np.ones((1, 8, 8), np.float32)) * 0.8 + 0.1 * rs.rand(10, 216, 256).astype(np.float32), 0, 1))
    aug = TripletAugmenter(vols, 160, 180, rs=np.random.RandomState(892372), device=device)
    B = 12
    trainer, pool = make_trainer("c2", device, B, 160, npool=1)
    host = [0.0]

    def loop(n, from_cache):
        for _ in range(n):
            if from_cache:
                t0 = time.perf_counter()
                batch = aug.next_batch(B, step=2, reuse_output=True)
                host[0] += time.perf_counter() - t0
            else:
                batch = pool[0]
            trainer.train(batch, keep_predictions=False)

    out = {}
    # host cost of a batch with an idle device (inside the loop the host runs ahead of the GPU until the launch queue pushes back, so
    # its per-call times there are waiting times)
    for _ in range(5):
        aug.next_batch(B, step=2, reuse_output=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        aug.next_batch(B, step=2, reuse_output=True)
    out["host_batch_us_idle_device"] = round(1e4 * (time.perf_counter() - t0), 1)
    torch.cuda.synchronize()
    for name, from_cache in (("resident_batch", False), ("volume_cache", True)):
        # the static inputs of the captured step: adopted from the first batch it is captured on -- capture on the kind it will run on
        trainer, pool = make_trainer("c2", device, B, 160, npool=1)
        loop(warmup, from_cache)
        torch.cuda.synchronize()
        host[0] = 0.0
        t0 = time.perf_counter()
        loop(steps, from_cache)
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[name] = {"ms_per_step": round(1e3 * dt / steps, 3), "slices_per_s": round(3 * B * steps / dt, 1),
                     "host_enqueue_ms_per_step": round(1e3 * t_enq / steps, 3)}
        if from_cache:
            out[name]["host_batch_us_in_loop"] = round(1e6 * host[0] / steps, 1)
        del trainer, pool
        torch.cuda.empty_cache()
    out["e2e_over_resident"] = round(out["volume_cache"]["slices_per_s"] / out["resident_batch"]["slices_per_s"], 4)
    out["reading"] = ("host_batch_us_idle_device = what the host spends on a batch (RandomState draws + descriptors + one launch) with nothing queued; "
                      "host_batch_us_in_loop and host_enqueue_ms_per_step are WAITING times: the host runs ahead of the device until the launch queue "
                      "pushes back, so inside the loop they approach the device's step time whatever the host cost is")
    # the budget of ONE RANK of the 8-rank run (rank 1 of 8: 2 of the 12 triplets): the host draws the random numbers of ALL 12 triplets (every
    # rank keeps the single-process stream position), assembles its own 2 and replays the 2-triplet captured step -- the host side of a batch
    # has to fit into a 0.85 ms step
    Br = 2
    for _ in range(5):
        aug.next_batch(B, step=2, reuse_output=True, shard=(1, 8))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        aug.next_batch(B, step=2, reuse_output=True, shard=(1, 8))
    rank = {"host_batch_us_idle_device": round(1e4 * (time.perf_counter() - t0), 1)}
    torch.cuda.synchronize()
    for name, from_cache in (("resident_batch", False), ("volume_cache", True)):
        trainer, pool = make_trainer("c2", device, Br, 160, npool=1)
        def rloop(n):
            for _ in range(n):
                trainer.train(aug.next_batch(B, step=2, reuse_output=True, shard=(1, 8)) if from_cache else pool[0], keep_predictions=False)
        rloop(warmup)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rloop(steps)
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rank[name] = {"ms_per_step": round(1e3 * dt / steps, 3), "host_enqueue_ms_per_step": round(1e3 * t_enq / steps, 3)}
        del trainer, pool
        torch.cuda.empty_cache()
    rank["e2e_over_resident"] = round(rank["resident_batch"]["ms_per_step"] / rank["volume_cache"]["ms_per_step"], 4)
    rank["workload"] = ("one rank of 8 (single process, no collectives): per step the draws of all 12 triplets, one assemble launch over its own 2, "
                        "the replayed 2-triplet step; %d steps" % steps)
    out["rank_of_8"] = rank
    out["workload"] = ("train_aesr.py's loop at configs[1]: %d steps, TripletAugmenter (host RandomState draws in the reference's order + one assemble "
                       "launch per batch) over a synthetic device-resident cache of 60 volumes 10x216x256 -> captured step" % steps)
    return out


def inference_bench(device):
    """BASELINE configs[4] inference leg: generate_hr_volumes.create_super_volume on a synthetic dHCP-shaped volume cropped to the
    evaluation patch (z = 30 slices of 224 x 224, 3 interpolations per pair), volume resident in HBM, output left in HBM."""
    import numpy as np
    from superresolution_aniso_mri_amd import generate_hr_volumes as ghv
    from superresolution_aniso_mri_amd.kwatsch.get_trainer import get_trainer_dynamic
    Z, H, n = 30, 224, 3
    torch.manual_seed(892372)
    tr = get_trainer_dynamic(build_args("c5", device), eval_mode=True)
    vol = torch.rand(Z, 1, H, H, device=device)
    alphas = np.linspace(0, 1, n + 2, endpoint=True)[1:-1]
    for _ in range(2):
        ghv.create_super_volume(tr, vol, alphas, use_original=True, to_cpu=False)
    torch.cuda.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        out = ghv.create_super_volume(tr, vol, alphas, use_original=True, to_cpu=False)["upsampled_image"]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    synth = (Z - 1) * n
    # algorithmic flops (direct convolutions of the reference's layers): every slice encoded once, every synthesised slice decoded once
    h1, h2, h4 = H + 2, (H + 2) // 2, (H + 2) // 4
    enc = 2.0 * (32 * h1 * h1 + 9 * (2 * 32 * 32 * h1 * h1 + (32 * 64 + 64 * 64) * h2 * h2 + (64 * 128 + 128 * 128) * h4 * h4))
    dec = 2.0 * 9 * ((128 * 64 + 64 * 64) * h4 * h4 + (64 * 32 + 32 * 32) * (2 * h4) ** 2 + (32 * 32 + 32) * (4 * h4) ** 2)
    gf = (Z * enc + synth * dec) / 1e9
    del tr
    torch.cuda.empty_cache()
    return {"workload": "generate_hr_volumes.create_super_volume: synthetic volume %d x %d x %d (eval_patch_size 224), %d interpolations per "
                        "slice pair, dHCP model (latent 128), random weights, input and output resident in HBM" % (Z, H, H, n),
            "ms_per_volume": round(dt * 1e3, 3), "synthesised_slices_per_s": round(synth / dt, 1), "output_slices": int(out.shape[0]),
            "algorithmic_gflop": round(gf, 2), "algorithmic_tflops": round(gf / dt / 1e3, 2),
            "algorithmic_frac_of_f32_mfma_peak": round(gf / dt / 1e3 / PEAK_F32_MFMA_TFLOPS, 4),
            "reference_executes": "2n(z-1) encoder passes + n(z-1) decoder passes (generate_hr_volumes.py:72-101); here z encoder passes, "
                                  "z decoder-stem passes and n(z-1) passes of the rest of the decoder"}


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
    ap.add_argument("--triplets", type=int, default=0, help="global batch in triplets (default: the BASELINE workload; other values are for experiments only)")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from the host instead of replaying the captured step")
    ap.add_argument("--dp-graph", choices=("on", "off"), default="on",
                    help="N > 1: 'on' (default) replays the step from captured graphs -- by default as graph SEGMENTS cut at the 9 collectives, "
                    "which stay eager enqueues on the same stream (RCCL, or host-staged gloo for rehearsals); AESR_DP_GRAPH=whole makes the "
                    "collectives nodes of ONE graph (RCCL only). 'off' launches every kernel from the host. A failure in either form ends the "
                    "run with a non-zero exit code: no other form is substituted")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary single-GPU measurements (c3, c3 with scales 3, c4, c5, "
                    "small shards, inference)")
    opt = ap.parse_args()

    from superresolution_aniso_mri_amd import engine
    from superresolution_aniso_mri_amd.parallel import DataParallelContext

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if opt.gpus > 1 and world != opt.gpus:
        raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`" % (opt.gpus, opt.gpus))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("AESR_SINGLE_DEVICE") == "1":       # rehearsal on a one-GPU box: every rank on cuda:0 (with AESR_DIST_BACKEND=gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = "cuda:%d" % local_rank
    dp = DataParallelContext(device=device)        # N > 1: an RCCL init failure raises with the RCCL error string (non-zero exit, no fallback)
    try:
        B, H = config_shape(opt.config, opt.triplets)
        use_graph = not opt.no_graph and (not dp.active or opt.dp_graph == "on")
        trainer, pool = make_trainer(opt.config, device, B, H, graph=use_graph, dp=dp)
        elapsed = timed_steps(trainer, pool, opt.steps, opt.warmup, dp)
        launch = "host launches"
        if use_graph:
            launch = ("captured HIP graph replay" if not dp.active else
                      "one HIP graph per step, RCCL collectives (library-owned communicator) as graph nodes" if dp.graph_mode == "whole" else
                      ("HIP graph segments between eager RCCL collectives (library-owned communicator)" if dp.data_backend == "rccl" else
                       "HIP graph segments between eager host-staged (gloo) collectives"))
        if dp.active:
            launch += "; graph form %s, data plane %s, SyncBN exchange %s" % (dp.graph_mode if use_graph else "none", dp.data_backend, dp.syncbn)
        if use_graph:
            if not getattr(trainer, "_graphs", None):
                raise SystemExit("bench: the step graph was requested but never captured")
        loss = trainer.losses["loss_ae"][-1]
        roofline = None
        if not opt.no_roofline and not dp.active:
            del trainer, pool
            trainer = pool = None
            torch.cuda.empty_cache()
            roofline = roofline_of(engine, device, max(1, min(opt.steps, 5)), opt.config, B, H, opt.gpus)
        if dp.rank != 0:
            return
        ms_per_step = 1e3 * elapsed / opt.steps
        line = {
            "metric": "training slices/sec (ae_combined, %dx%d, latent=128)" % (H, H), "value": round(3 * B * opt.steps / elapsed, 1),
            "unit": "slices/s", "n_gpus": opt.gpus, "steps": opt.steps, "warmup": opt.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": WORKLOAD[opt.config], "global_batch_triplets": B, "slices_per_step": 3 * B,
                       "parallelism": "dp%d" % opt.gpus, "init": "reference Initializer, seed 892372, random weights", "launch": launch},
            "final_loss": round(float(loss), 6),
        }
        line.update(step_numbers(opt.config, B, H, ms_per_step, engine.FUSE_STEM))
        if roofline is not None:
            line["roofline"] = roofline
        del trainer, pool
        torch.cuda.empty_cache()
        if opt.gpus == 1 and not opt.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(opt.config, B, H)
        if opt.gpus == 1 and opt.config == "c2" and B == 12 and not opt.no_secondary:
            sec = {}
            sec["c3"] = secondary_config(engine, "c3", device, with_roofline=not opt.no_roofline, with_cpu=not opt.no_cpu_baseline)
            sec["c3_scales3"] = secondary_config(engine, "c3", device, scales3=True)
            sec["c4"] = secondary_config(engine, "c4", device, steps=6, warmup=3)
            sec["c5"] = secondary_config(engine, "c5", device, steps=6, warmup=3)
            sec["small_shards"] = small_shards(device)
            sec["inference"] = inference_bench(device)
            sec["e2e"] = e2e_bench(device)
            line["secondary"] = sec
        # kernel sources of this run (the hash the PMC files are matched against) and the ring kernel's protocol watchdog (0 = no wait
        # on an LDS arrival counter ever gave up in this process)
        from superresolution_aniso_mri_amd._hip import lib as _lib
        line["csrc_sha"] = csrc_sha()
        line["ring_watchdog_timeouts"] = int(_lib.aesr_conv2d_wino_ring_timeouts())
        line["bn_barrier_timeouts"] = int(_lib.aesr_bn_fused1_timeouts())
        if line["ring_watchdog_timeouts"] or line["bn_barrier_timeouts"]:          # a number measured on garbage is not a number: no line, non-zero exit
            raise SystemExit("bench: kernel-side watchdogs fired (ring arrival counters %d, BatchNorm grid barrier / peer exchange %d); no result line "
                             "is printed" % (line["ring_watchdog_timeouts"], line["bn_barrier_timeouts"]))
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    finally:
        dp.shutdown()


if __name__ == "__main__":
    main()
