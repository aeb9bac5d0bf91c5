#!/usr/bin/env python
"""Where does the HIP step DECIDE differently from exact arithmetic, and what is left of the gradient differences once it is asked the same
question?  (Test infrastructure: imports oracle/, not collected by pytest.)

    python tests/diag_routing.py [small] [c2] [c3] > profiles/rNN_routing_report.txt

For every golden step fixture (``small``) and for BASELINE configs[1] / configs[2] at full size (``c2`` / ``c3``): one eager HIP step with
its non-smooth decisions recorded (tests/routing_util.py), the oracle evaluated in fp64 with its own decisions and with the HIP
path's; printed: the decisions that differ with their fp64 margins (a margin of ~1e-7 of the layer's rms = a tie within fp32 rounding),
and first-step gradient rel-L2 HIP vs fp64[own decisions] / fp64[HIP decisions] (worst tensor, median, whole gradient)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import routing_util as ru  # noqa: E402
from oracle import ae_oracle, lpips_oracle, routing, step_oracle  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-300))


def lpips_kw():
    lin = np.load(os.path.join(ROOT, "superresolution_aniso_mri_amd", "lpips", "weights", "v0.1", "vgg_lin.npz"))
    return dict(vgg_sd=lpips_oracle.hash_vgg16_state(), lin_w=[torch.from_numpy(lin["lin%d" % k]).reshape(1, -1, 1, 1) for k in range(5)])


def summarize(title, g_hip, g_ref):
    errs = sorted((rel(g_hip[k], g_ref[k]), k) for k in g_ref)
    num = sum(float((g_hip[k].double().cpu() - g_ref[k].double()).pow(2).sum()) for k in g_ref)
    den = sum(float(g_ref[k].double().pow(2).sum()) for k in g_ref)
    print("    %-44s worst %.2e (%s)  median %.2e  whole gradient %.2e" % (title, errs[-1][0], errs[-1][1], errs[len(errs) // 2][0], (num / den) ** 0.5))


def report(name, trainer, make_ost, batch, fixture_grads=None):
    t0 = time.time()
    dec = ru.hip_step_decisions(trainer, batch)
    g_hip = {k: p.grad.detach().clone() for k, p in trainer.model.named_parameters()}
    r_own, g_own, _ = ru.oracle64_step(make_ost, batch)
    diffs = routing.differing_decisions(r_own, dec)
    ntot = sum(v.numel() for v in dec.values())
    print("%s: %d decisions recorded, %d differ from the oracle's fp64 evaluation  (%.0f s)" % (name, ntot, len(diffs), time.time() - t0))
    if diffs:
        print(ru.describe(diffs, 16))
    summarize("HIP vs fp64 oracle, its own decisions", g_hip, g_own)
    if diffs:
        _, g_f, _ = ru.oracle64_step(make_ost, batch, forced=dec)
        summarize("HIP vs fp64 oracle, the HIP path's decisions", g_hip, g_f)
    if fixture_grads is not None:
        summarize("HIP vs the reference fixture (fp32 CPU)", g_hip, fixture_grads)
        summarize("reference fixture vs fp64 oracle, own decisions", fixture_grads, g_own)
    sys.stdout.flush()


def small():
    import test_gpu_step as tgs
    import test_oracle_golden as tog
    for tag in sorted(tgs.STEP_CASES):
        rec = dict(np.load(os.path.join(GOLDEN, "step_k3_%s.npz" % tag)))
        kw, lr, _ = tog.STEP_CASES[tag]

        def make_ost(rec=rec, kw=kw, lr=lr, tag=tag):
            ae = ae_oracle.OracleAE(tog.small_cfg(tag), init=False).load_state_dict({k[3:]: torch.from_numpy(v) for k, v in rec.items() if k.startswith("p0/")})
            return step_oracle.OracleStep(ae, lr=lr, ex_loss_weight1=0.05, **lpips_kw(), **kw)
        if kw.get("plain"):
            continue                          # plain ``ae``: one pass pair, no synthesis branch -- its names differ; MSE only, no ties seen
        trainer = tgs.make_trainer(tag, rec)
        fx = {k[6:]: torch.from_numpy(v) for k, v in rec.items() if k.startswith("grad0/")}
        report("step_k3_" + tag, trainer, make_ost, tgs._batch(rec, 0), fx)


def full(tag):
    import test_gpu_baseline_parity as tbp
    from superresolution_aniso_mri_amd.data_synth import synthetic_batch
    loss = "mse" if tag == "c2" else "perceptual"
    for seed in (892372, 7):
        tr, _, _ = tbp._pair(loss, seed=seed)
        sd = {k: v.detach().cpu().clone() for k, v in tr.model.state_dict().items()}

        def make_ost(sd=sd):
            cfg = dict(width=128, latent_width=32, depth=32, latent=128, colors=1, use_batchnorm=True, use_sigmoid=True)
            oracle = ae_oracle.OracleAE(cfg, init=False).load_state_dict(sd)
            return step_oracle.OracleStep(oracle, lr=1e-5, ex_loss_weight1=0.05, image_mix_loss_func=loss, **(lpips_kw() if loss == "perceptual" else {}))
        report("%s (12 triplets 160x160), seed %d" % (tag, seed), tr, make_ost, synthetic_batch(12, 160, 160, seed=seed))
        del tr
        torch.cuda.empty_cache()


if __name__ == "__main__":
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    what = sys.argv[1:] or ["small"]
    print("# non-smooth decisions of the HIP step against the oracle in fp64 (tests/diag_routing.py %s)" % " ".join(what))
    if "small" in what:
        small()
    for tag in ("c2", "c3"):
        if tag in what:
            full(tag)
