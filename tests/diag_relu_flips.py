#!/usr/bin/env python
"""Why first-step gradients at BASELINE config C2 differ by 1-2e-4 between the HIP path and the CPU oracle although every
forward quantity agrees to <1e-6 (test infrastructure: imports oracle/, not collected by pytest).

    python tests/diag_relu_flips.py > profiles/rNN_gradient_flip_analysis.txt

1. decoder backward at IDENTICAL latents: HIP vs the oracle in fp64, next to the CPU fp32 oracle vs fp64;
2. the fp64 decoder's gradients at three latents -- exact (fp64 encoder), CPU-fp32 encoder, HIP encoder;
3. LeakyReLU sign flips between those latents in the decoder's first two convolutions, and what ONE flip does."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import ae_oracle, step_oracle  # noqa: E402
from superresolution_aniso_mri_amd import ops  # noqa: E402
from superresolution_aniso_mri_amd.data_synth import synthetic_batch  # noqa: E402
from superresolution_aniso_mri_amd.kwatsch.get_trainer import get_trainer_dynamic  # noqa: E402


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-300))


def as_fp64(o):
    o.params = type(o.params)((k, v.detach().double().requires_grad_(True)) for k, v in o.params.items())
    o.buffers = type(o.buffers)((k, v.double() if v.is_floating_point() else v) for k, v in o.buffers.items())
    return o


torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
torch.manual_seed(892372)
tr = get_trainer_dynamic(bench.build_args("c2", "cuda:0"))
cfg = dict(width=128, latent_width=32, depth=32, latent=128, colors=1, use_batchnorm=True, use_sigmoid=True)
sd = {k: v.detach().cpu() for k, v in tr.model.state_dict().items()}
o64 = as_fp64(ae_oracle.OracleAE(cfg, init=False).load_state_dict(sd))
B, lam = 12, 0.05
batch = synthetic_batch(B, 160, 160, seed=892372)
x, btw = batch["image"], batch["slice_between"]
m = tr.model
m.train()
with torch.no_grad():
    z_hip = m.encode(x.cuda()).detach()


def oracle_step(dt):
    oo = ae_oracle.OracleAE(cfg, init=False).load_state_dict(sd)
    if dt == torch.float64:
        as_fp64(oo)
    r = step_oracle.OracleStep(oo, lr=1e-5, ex_loss_weight1=lam, image_mix_loss_func="mse").train(x.to(dt), btw.to(dt))
    return {k: p.grad.detach().clone() for k, p in oo.params.items()}, r["z"].detach()


def decoder_grads(oo, zin, dt):
    zc = zin.detach().cpu().to(dt).clone().requires_grad_(True)
    zm = 0.5 * zc[:B] + 0.5 * zc[B:]
    for p in oo.params.values():
        p.grad = None
    loss = F.mse_loss(oo.decode(zc, train=True), x.to(dt)) + lam * F.mse_loss(oo.decode(zm, train=True), btw.to(dt))
    loss.backward()
    return {k: p.grad.detach().clone() for k, p in oo.params.items() if k.startswith("dec.")}


print("C2 (12 triplets 160x160, MSE synthesis loss, seed 892372): where do first-step gradient differences of 1-2e-4 come from?")
z = z_hip.clone().requires_grad_(True)
out, smix = m.decode_multi([z, ops.lerp_mix(z, 0.5, 0.5)])
m.zero_grad()
(F.mse_loss(out, x.cuda()) + lam * F.mse_loss(smix, btw.cuda())).backward()
g_hip = {k: p.grad.detach().clone() for k, p in m.named_parameters() if k.startswith("dec.")}
g64_at_hip = decoder_grads(o64, z_hip, torch.float64)
g32_at_hip = decoder_grads(ae_oracle.OracleAE(cfg, init=False).load_state_dict(sd), z_hip, torch.float32)
print("1. decoder backward at the SAME latents (those of the HIP encoder), rel-L2 against the oracle evaluated in fp64:")
for k in g_hip:
    print("     %-14s HIP %.2e   CPU fp32 oracle %.2e" % (k, rel(g_hip[k], g64_at_hip[k]), rel(g32_at_hip[k], g64_at_hip[k])))
g64_full, z64 = oracle_step(torch.float64)
_, z32 = oracle_step(torch.float32)
g_at = {"exact latents": decoder_grads(o64, z64, torch.float64), "CPU-fp32 latents": decoder_grads(o64, z32, torch.float64),
        "HIP latents": g64_at_hip}
print("2. the fp64 decoder at three sets of latents (rel-L2 of the latents to the exact ones: CPU fp32 %.2e, HIP %.2e):"
      % (rel(z32, z64), rel(z_hip, z64)))
for k in ("dec.0.weight", "dec.2.weight", "dec.2.bias", "dec.6.weight", "dec.12.weight"):
    print("     %-14s gradient change vs exact latents: CPU-fp32 latents %.2e   HIP latents %.2e"
          % (k, rel(g_at["CPU-fp32 latents"][k], g_at["exact latents"][k]), rel(g_at["HIP latents"][k], g_at["exact latents"][k])))


def preacts(zin):
    with torch.no_grad():
        zc, P = zin.detach().cpu().double(), o64.params
        a0 = F.conv2d(zc, P["dec.0.weight"], P["dec.0.bias"], padding=1)
        return a0, F.conv2d(F.leaky_relu(a0, 0.01), P["dec.2.weight"], P["dec.2.bias"], padding=1)


print("3. LeakyReLU inputs of dec.0 / dec.2 whose SIGN differs from the exact evaluation (the derivative there jumps 0.01 <-> 1):")
p64 = preacts(z64)
d = z_hip.cpu().double() - z64
for name, zz in (("HIP latents", z_hip), ("CPU-fp32 latents", z32), ("exact - (HIP - exact)", z64 - d)):
    pp = preacts(zz)
    for lname, a, b in (("dec.0", pp[0], p64[0]), ("dec.2", pp[1], p64[1])):
        flips = ((a > 0) != (b > 0))
        idx = flips.nonzero()
        print("     %-22s %s: %d of %d%s" % (name, lname, int(flips.sum()), flips.numel(), "" if not len(idx) else
              "   (image %d: exact value %.2e, here %.2e; rms of the layer %.2e)"
              % (int(idx[0][0]), float(b[tuple(idx[0])]), float(a[tuple(idx[0])]), float(b.pow(2).mean().sqrt()))))
half = d.clone()
half[B:] = 0
ga = g_at["exact latents"]
for name, comp in (("images 0..11 of (HIP - exact)", half), ("images 12..23", d - half), ("-(HIP - exact)", -d)):
    g = decoder_grads(o64, z64 + comp, torch.float64)
    print("     exact latents + %-30s dec.0.weight %.2e   dec.2.weight %.2e" % (name, rel(g["dec.0.weight"], ga["dec.0.weight"]),
                                                                                  rel(g["dec.2.weight"], ga["dec.2.weight"])))
print("Reading: at equal inputs the HIP backward is as close to fp64 as (here: closer than) the CPU fp32 path.  The 1-2e-4 of the full\n"
      "step is ONE LeakyReLU input that is ~1e-9 in exact arithmetic (4e-6 of the layer's rms): rounding puts it on the other side of\n"
      "zero, its derivative flips, and every gradient upstream of that layer moves by ~1e-4 -- not linearly in the perturbation (the\n"
      "mirrored perturbation flips two OTHER elements and moves the gradients by 2e-5).  Any two fp32 implementations can differ this\n"
      "way; Adam bounds the effect on a step by lr.")
