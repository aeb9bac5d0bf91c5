"""Test infrastructure (imports oracle/): one eager ``ae_combined`` step of the HIP trainer with every non-smooth decision it took
recorded -- LeakyReLU signs of the auto-encoder passes, ReLU signs and max-pool winners of the LPIPS-VGG stack -- under the names the
routing-aware oracle uses (oracle/routing.py, oracle/step_oracle.OracleStep.train(route=...)), and the fp64 evaluations around it.

Names: ``x/enc.<i>`` (encoder pass over the 2B input slices; i = nn.Sequential index of the LeakyReLU), ``z/dec.<i>`` (decoder pass over
their latents), ``mix/dec.<i>`` (decoder pass over the B mixes), ``lp_rec/in1/relu<n>|pool<n>`` (LPIPS as the reconstruction loss:
the reconstruction is the ``pred`` argument), ``lp_syn/in0/...`` (LPIPS synthesis loss: the synthesised slice is the ``target``
argument) -- n = 1-based VGG conv count."""
import torch

from oracle import routing


def nchw(t):
    return t.detach().permute(0, 3, 1, 2).cpu()


def hip_step_decisions(trainer, batch):
    """Runs ``trainer.train(batch)`` (host-launched) and returns {decision name: bool mask | winner index} in the oracle's layout."""
    from superresolution_aniso_mri_amd import _hip, engine
    from superresolution_aniso_mri_amd.lpips import networks_basic as nb
    B = batch["image"].shape[0] // 2
    engine.TRACE, nb._TRACE = [], []
    try:
        trainer.train(batch, keep_predictions=True)
        torch.cuda.synchronize()
        etrace, ltrace = engine.TRACE, nb._TRACE
    finally:
        engine.TRACE, nb._TRACE = None, None
    model = trainer.model
    dec = {}
    for seq, mod, act, out in etrace:
        if act != _hip.ACT_LRELU:
            continue
        i = list(seq).index(mod) + 1                      # the LeakyReLU behind the convolution
        if seq is model.enc:
            assert out.shape[0] == 3 * B, out.shape       # [x (2B) | slice_between (B)]: the second group is logging-only
            dec["x/enc.%d" % i] = nchw(out[:2 * B]) > 0
        elif seq is model.dec:
            assert out.shape[0] == 3 * B, out.shape       # [z (2B) | z_mix (B)]
            dec["z/dec.%d" % i] = nchw(out[:2 * B]) > 0
            dec["mix/dec.%d" % i] = nchw(out[2 * B:]) > 0
    for kind, acts in [t for t in ltrace if t[0] == "acts"]:
        n2 = acts[0].shape[0]
        nb_ = n2 // 2                                      # both branches in one batch; the differentiated branch comes first
        tag = {2 * B: "lp_rec/in1/", B: "lp_syn/in0/"}[nb_]
        for n, a in enumerate(acts, start=1):
            a0 = nchw(a[:nb_])
            dec[tag + "relu%d" % n] = a0 > 0
            if n in nb.TAP_AFTER_CONV[:-1]:
                dec[tag + "pool%d" % n] = routing.Routing.windows(a0).argmax(-1)
    return dec


def as_fp64(oracle_ae):
    oracle_ae.params = type(oracle_ae.params)((k, v.detach().double().requires_grad_(True)) for k, v in oracle_ae.params.items())
    oracle_ae.buffers = type(oracle_ae.buffers)((k, v.double() if v.is_floating_point() else v.clone()) for k, v in oracle_ae.buffers.items())
    return oracle_ae


def oracle64_step(make_oracle_step, batch, forced=None):
    """One fp64 evaluation of the oracle step (no parameter update).  ``make_oracle_step()`` -> a fresh OracleStep on fp32 parameters.
    Returns (route, {param name: gradient}, result dict)."""
    ost = make_oracle_step()
    as_fp64(ost.ae)
    if ost.vgg_sd is not None:
        ost.vgg_sd = {k: v.double() for k, v in ost.vgg_sd.items()}
        ost.lin_w = [w.double() for w in ost.lin_w]
    ost.opt = torch.optim.SGD(ost.ae.parameters(), lr=0.0)
    route = routing.Routing(forced)
    kw = {}
    if "alpha_from" in batch:
        kw = dict(alpha_from=batch["alpha_from"].double(), alpha_to=batch["alpha_to"].double())
    res = ost.train(batch["image"].double(), batch["slice_between"].double(), route=route, **kw)
    return route, {k: p.grad.detach().clone() for k, p in ost.ae.params.items()}, res


def describe(diffs, limit=12):
    lines = []
    for d in diffs[:limit]:
        lines.append("  %-22s %-4s at %-18s margin %.3e = %.2e of the layer's rms %.3e" % (d["name"], d["kind"], d["index"], d["margin"], d["rel"], d["scale"]))
    if len(diffs) > limit:
        lines.append("  ... and %d more" % (len(diffs) - limit))
    return "\n".join(lines)
