"""Background MLP (second half): the stand-in module against golden vectors made by the reference's
own BackgroundModelSimple (tests/golden/gen_bgmlp.py) -- on CPU in float32 (exact formula, channel order), and on the
GPU through the fused MFMA kernels (bf16 operands, fp32 accumulation: tolerances stated below)."""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT

GOLDEN = os.path.join(ROOT, "tests", "golden", "bgmlp.npz")


def _load(device, fused=True):
    import __graft_entry__  # noqa: F401
    from ava256_amd.trainloop import BackgroundMLPStandIn
    g = np.load(GOLDEN)
    m = BackgroundMLPStandIn(3, 2, fused=fused)
    sd = {}
    for k in g.files:
        if k.startswith("param/") and "ident" not in k:
            v = torch.from_numpy(g[k])
            sd[k[6:]] = v.reshape(v.shape[0], v.shape[1]) if v.dim() == 4 else v    # 1x1 Conv2d -> Linear
    missing = m.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return m.to(device), g


def _run(m, g, device):
    cam, idx = torch.from_numpy(g["camindex"]).to(device), torch.from_numpy(g["idindex"]).to(device)
    sc = torch.from_numpy(g["samplecoords"]).to(device)
    bg = m(cam, idx, sc)
    (bg * torch.from_numpy(g["gout"]).to(device)).sum().backward()
    grads = {k: p.grad.detach().cpu().numpy() for k, p in m.named_parameters()}
    return bg.detach().cpu().numpy(), grads


def _cos(a, b):
    a, b = a.ravel().astype(np.float64), b.ravel().astype(np.float64)
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))


def test_standin_matches_the_reference_module_on_cpu():
    m, g = _load("cpu")
    bg, grads = _run(m, g, "cpu")
    assert np.abs(bg - g["bg"]).max() <= 1e-4 * np.abs(g["bg"]).max()
    for k, v in grads.items():
        ref = g["grad/" + k].reshape(v.shape)
        assert np.abs(v - ref).max() <= 1e-4 * np.abs(ref).max(), k


@pytest.mark.gpu
def test_fused_kernels_match_the_reference_module():
    """bf16 operands and stored activations: the output (mean 100, spread from the 25x scale) is held to 2 % of its
    spread.  Gradients: bf16 rounding of a pre-activation near zero flips its LeakyReLU slope (1 <-> 0.2), so against
    the float32 reference ANY bf16 execution of this network sits near 10 % norm-wise in the early layers (eager
    autocast: cosine 0.993 / 12 %, these kernels: 0.996 / 9 %, gpurun_out/r02y); held to cosine >= 0.99 and 15 %, and
    in the test below to no worse than eager bf16 autocast."""
    m, g = _load("cuda")
    bg, grads = _run(m, g, "cuda")
    spread = np.abs(g["bg"] - 100.0).max()
    assert np.abs(bg - g["bg"]).max() <= 2e-2 * spread, (np.abs(bg - g["bg"]).max(), spread)
    for k, v in grads.items():
        ref = g["grad/" + k].reshape(v.shape)
        assert _cos(v, ref) >= 0.99, (k, _cos(v, ref))
        assert np.linalg.norm(v - ref) <= 0.15 * np.linalg.norm(ref), (k, np.linalg.norm(v - ref) / np.linalg.norm(ref))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 96, 80), (1, 37, 53), (3, 128, 128)])
def test_fused_kernels_match_eager_fp32_on_ragged_images(shape):
    """Image sizes that are not multiples of the 128-pixel tile; the stand-in's own seeded weights; eager float32 on
    the same device as the reference, and eager bf16 autocast as the yardstick of what the dtype costs: the fused
    kernels may not be further from float32 than 1.25 x the eager bf16 run (+ 1 %)."""
    import __graft_entry__  # noqa: F401
    from ava256_amd.trainloop import BackgroundMLPStandIn
    B, H, W = shape
    gen = torch.Generator().manual_seed(B * 1000 + H)
    cam, idx = torch.randint(0, 5, (B,), generator=gen).cuda(), torch.randint(0, 3, (B,), generator=gen).cuda()
    sc = (torch.rand(B, H, W, 2, generator=gen) * 2 - 1).cuda()
    gout = torch.randn(B, 3, H, W, generator=gen).cuda()
    res = []
    for fused, dt in ((True, torch.bfloat16), (False, None), (False, torch.bfloat16)):
        m = BackgroundMLPStandIn(5, 3, autocast_dtype=dt, fused=fused).cuda()
        with torch.no_grad():
            for p in m.parameters():
                if p.dim() == 1:
                    p.copy_(0.1 * torch.randn(p.shape, generator=torch.Generator().manual_seed(p.numel())).cuda())
        bg = m(cam, idx, sc)
        (bg * gout).sum().backward()
        res.append((bg.detach().cpu().numpy(), {k: p.grad.detach().cpu().numpy() for k, p in m.named_parameters()}))
    (bg1, g1), (bg0, g0), (bg2, g2) = res
    spread = np.abs(bg0 - 100.0).max()
    assert np.abs(bg1 - bg0).max() <= 2e-2 * spread, (np.abs(bg1 - bg0).max(), spread)
    for k in g0:
        n0 = np.linalg.norm(g0[k])
        mine, eager = np.linalg.norm(g1[k] - g0[k]) / n0, np.linalg.norm(g2[k] - g0[k]) / n0
        assert _cos(g1[k], g0[k]) >= 0.99, (k, _cos(g1[k], g0[k]))
        assert mine <= 1.25 * eager + 1e-2, (k, mine, eager)


@pytest.mark.gpu
def test_inference_path_equals_training_path():
    """Without gradients nothing is kept for a backward (acts / x0 are NULL in the C ABI): same kernel, same output bits."""
    m, g = _load("cuda")
    cam, idx = torch.from_numpy(g["camindex"]).cuda(), torch.from_numpy(g["idindex"]).cuda()
    sc = torch.from_numpy(g["samplecoords"]).cuda()
    with torch.no_grad():
        a = m(cam, idx, sc)
    b = m(cam, idx, sc)
    assert b.requires_grad and not a.requires_grad
    assert torch.equal(a, b.detach())
