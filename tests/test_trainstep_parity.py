"""ONE training iteration on the GPU against a float64 replay.

`Trainer.step` (ddp-train.py:362-442: forward with the schedule of the iteration, the four loss terms and their
weights, backward, NaN/Inf mask, clip, Adam) runs once on the GPU through the gfx950 kernels and once on the CPU in
float64 with the SAME modules, where the renderer is an autograd Function whose forward / backward are the float64
oracle (oracle/mvp_oracle.c: pinned against the reference's dense autograd statement, tests/test_oracle_golden.py).
The loss formula both sides evaluate is pinned to the reference's own statements by tests/golden/trainstep_loss.npz
(tests/test_trainloop.py::test_loss_formula_is_the_references_own).  Compared: the loss and each of its terms, the
gradient norm before clipping, every parameter's (clipped) gradient, the parameters after the Adam step, the
`adaptwarps` buffer.  The background MLP runs in eager float32 here so that bf16 is not part of the comparison
(its kernels have their own fixtures, tests/test_bgmlp.py)."""
import copy

import numpy as np
import pytest
import torch

import __graft_entry__  # noqa: F401  (repo root on sys.path)

VOLRADIUS = 256.0


class _OracleMarch(torch.autograd.Function):
    """mvpraymarch with the float64 oracle as forward and backward (test infrastructure)."""

    @staticmethod
    def forward(ctx, primpos, primrot, primscale, template, rays, stepsize, oracle):
        rp, rd, tm = rays
        a = (rp, rd, stepsize, tm, primpos.detach().numpy(), primrot.detach().numpy(), primscale.detach().numpy(),
             template.detach().numpy())
        rgba, raysat, st = oracle.march_forward(*a)
        ctx.a, ctx.raysat, ctx.oracle = a, raysat, oracle
        _OracleMarch.last_stats = st
        return torch.from_numpy(rgba)

    @staticmethod
    def backward(ctx, g):
        gp, gr, gs, gt = ctx.oracle.march_backward(*ctx.a, ctx.raysat, np.ascontiguousarray(g.numpy()))
        return torch.from_numpy(gp), torch.from_numpy(gr), torch.from_numpy(gs), torch.from_numpy(gt), None, None, None


def oracle_renderer(oracle):
    def render(camrot, campos, focal, princpt, pixelcoords, decout):
        rays = oracle.raydirs(campos.numpy(), camrot.numpy(), focal.numpy(), princpt.numpy(), pixelcoords.numpy(), VOLRADIUS)
        rgba = _OracleMarch.apply(decout["primpos"], decout["primrot"], decout["primscale"], decout["template"], rays,
                                  1.0 / VOLRADIUS, oracle)
        nchw = rgba.permute(0, 3, 1, 2)
        return nchw[:, :3].contiguous(), nchw[:, 3:4].contiguous()
    return render


def build_model(K, alpha_init, renderer=None):
    """The full decode path of the training loop with seeded, non-trivial parameters (float32 construction: the float64
    twin is an exact copy of these numbers)."""
    from ava256_amd.trainloop import (BackgroundMLPStandIn, CodeEncoderStandIn, ColorCalStandIn, RaymarchTrainModel,
                                      SlabDecoderStandIn)
    model = RaymarchTrainModel(SlabDecoderStandIn(K, seed=1, alpha_init=alpha_init), VOLRADIUS, renderer=renderer,
                               colorcal=ColorCalStandIn(8, 2), encoder=CodeEncoderStandIn(),
                               bgmodel=BackgroundMLPStandIn(8, 2, autocast_dtype=None, fused=False))
    g = torch.Generator().manual_seed(99)
    with torch.no_grad():  # off the identity / zero initialisations, so that every term of the forward matters
        for name, p in model.named_parameters():
            if name.startswith("colorcal") or name in ("decoder.pos_delta", "decoder.rotvec", "decoder.logscale"):
                p.add_(0.1 * torch.randn(p.shape, generator=g))
            elif p.dim() == 1:
                p.add_(0.05 * torch.randn(p.shape, generator=g))
    return model


def replay_step(model32, batch_cpu, iternum, oracle, lr):
    """The same step in float64 on the CPU; returns (trainer, loss, terms)."""
    from ava256_amd.trainloop import Trainer, forward_schedule
    m = copy.deepcopy(model32).cpu().double()
    m._renderer = oracle_renderer(oracle)
    b = {k: (v.double() if v.is_floating_point() else v) for k, v in batch_cpu.items()}
    if iternum >= 100:  # the running average was filled during the first 100 iterations
        with torch.no_grad():
            m.decoder(m.encoder(b["code"], b["noise"])[0], schedule=forward_schedule(0), gt_geo=b["verts"])
    tr = Trainer(m, lr=lr)
    tr.iternum = iternum
    loss, terms = tr.step(b)
    return tr, float(loss), {k: float(v) for k, v in terms.items()}


def _cpu_batch(N, H, W, K, seed=3):
    """Batch keys of make_training_batch with a synthetic target image (CPU: no kernels to render a target with)."""
    from ava256_amd.trainloop import make_training_batch
    b, _ = make_training_batch(N, H, W, K, "cpu", seed=seed, ncams=8, nident=2)
    g = torch.Generator().manual_seed(seed)
    b["image"] = 60.0 + 30.0 * torch.rand(N, 3, H, W, generator=g)
    return b


@pytest.mark.parametrize("iternum", [0, 100])
def test_float64_replay_runs_on_cpu(oracle64, iternum):
    """The checker side alone (small): executes, every loss term is present and finite, gradients reach every parameter
    group the schedule leaves active, Adam moves the parameters."""
    K = 16
    model = build_model(K, alpha_init=6.0)
    before = {n: p.detach().clone().double() for n, p in model.named_parameters()}
    tr, loss, terms = replay_step(model, _cpu_batch(1, 16, 16, K), iternum, oracle64, lr=1e-3)
    assert sorted(terms) == ["irgbl1", "kldiv", "primvolsum", "vertl1"] and np.isfinite(loss)
    assert _OracleMarch.last_stats["rays_hit"] > 0
    named = dict(tr.raw_model.named_parameters())
    for n in ("decoder.tex", "decoder.opacity", "decoder.gain.weight", "decoder.geo_head.weight", "colorcal.wcam",
              "encoder.mu.weight", "bgmodel.mlp.0.weight"):
        assert float(named[n].grad.abs().max()) > 0.0, n
    for n in ("decoder.pos_delta", "decoder.rotvec", "decoder.logscale"):   # residuals_weight = 0 switches them off
        assert (float(named[n].grad.abs().max()) > 0.0) == (iternum >= 100), n
    assert any(not torch.equal(named[n].detach(), before[n]) for n in named)
    assert float(tr.raw_model.decoder.adaptwarps.min()) > 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("iternum", [0, 100])
def test_one_training_iteration_matches_the_float64_replay(oracle64, iternum):
    from ava256_amd.trainloop import SlabDecoderStandIn, Trainer, forward_schedule, make_training_batch
    dev = "cuda"
    N, H, W, K, lr = 2, 64, 64, 256, 1.0e-3
    alpha_init = 6.0   # about half of the hitting rays saturate (asserted below from the oracle's own count)
    model = build_model(K, alpha_init)
    batch, _ = make_training_batch(N, H, W, K, dev, seed=3, ncams=8, nident=2,
                                   target_decoder=SlabDecoderStandIn(K, seed=9, alpha_init=alpha_init))
    batch_cpu = {k: v.detach().cpu() for k, v in batch.items()}

    rt, rloss, rterms = replay_step(model, batch_cpu, iternum, oracle64, lr)
    st = _OracleMarch.last_stats
    frac_sat = st["rays_saturated"] / max(1, st["rays_hit"])
    assert 0.25 <= frac_sat <= 0.75, frac_sat

    gm = copy.deepcopy(model).to(dev)
    if iternum >= 100:
        with torch.no_grad():
            gm.decoder(gm.encoder(batch["code"], batch["noise"])[0], schedule=forward_schedule(0), gt_geo=batch["verts"])
    gt_ = Trainer(gm, lr=lr)
    gt_.iternum = iternum
    before = {n: p.detach().clone() for n, p in gm.named_parameters()}
    gloss, gterms = gt_.step(batch)
    torch.cuda.synchronize()

    # ---- loss and its terms (fp32 kernels and fp32 eager modules against float64) ----
    assert abs(float(gloss) - rloss) <= 2e-4 * abs(rloss), (float(gloss), rloss)
    for k, v in rterms.items():
        assert abs(float(gterms[k]) - v) <= 2e-4 * max(abs(v), 1e-3 * abs(rloss)), (k, float(gterms[k]), v)
    # ---- gradient norm before clipping (what clip_grad_norm_ returns) ----
    gn, rn = float(gt_.last_grad_norm), float(rt.last_grad_norm)
    assert rn > 1.0, rn                                     # so the clip really scales the gradients
    assert abs(gn - rn) <= 1e-3 * rn, (gn, rn)
    # ---- every parameter: clipped gradient, and the parameter after the Adam step ----
    rnamed = dict(rt.raw_model.named_parameters())
    worst = {}
    for n, p in gm.named_parameters():
        g, r = p.grad.detach().double().cpu(), rnamed[n].grad.detach()
        rnorm = float(r.norm())
        if rnorm == 0.0:                                    # switched off by the schedule (residuals_weight = 0)
            assert float(g.abs().max()) == 0.0, n
            continue
        cos = float((g * r).sum() / (g.norm() * r.norm()))
        rel = float((g - r).norm()) / rnorm
        worst[n] = (cos, rel)
        assert cos >= 0.9999, (n, cos, rel)
        assert rel <= 1.5e-2, (n, cos, rel)
        # Adam, first step (m = 0.1 g, v = 0.001 g^2, both bias-corrected): update = -lr * g / (|g| + eps), eps = 1e-8.
        # (a) the GPU step applies exactly that to ITS OWN clipped gradient;  (b) against the replay wherever the
        # gradient is well above eps and its sign beyond doubt (below that the update is a steep function of g: a
        # clipped gradient of 1e-8 turns a 1e-9 difference into 2.5 % of lr)
        dp_g = (p.detach() - before[n]).double().cpu()
        dp_r = rnamed[n].detach() - before[n].double().cpu()
        ulp = float(np.spacing(np.float32(float(before[n].abs().max()))))
        own = -lr * g / (g.abs() + 1e-8)
        assert float((dp_g - own).abs().max()) <= 1e-3 * lr + 2.0 * ulp, n
        clear = (r.abs() > 1e-2 * r.abs().max()) & (r.abs() > 1e-5)
        if bool(clear.any()):
            assert float((dp_g - dp_r)[clear].abs().max()) <= 2e-2 * lr + 2.0 * ulp, n
    assert len(worst) >= 20
    print("iteration", iternum, "saturated", round(frac_sat, 3), "worst cos", min(v[0] for v in worst.values()),
          "worst norm-wise", max(v[1] for v in worst.values()), "grad norm", gn, rn, "loss", float(gloss), rloss)
    # ---- the running average of the primitive sizes ----
    aw_g, aw_r = gm.decoder.adaptwarps.double().cpu(), rt.raw_model.decoder.adaptwarps
    assert float((aw_g - aw_r).abs().max()) <= 1e-5 * float(aw_r.abs().max())
