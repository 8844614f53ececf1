"""Background MLP (second half): the stand-in module against golden vectors made by the reference's
own BackgroundModelSimple (tests/golden/gen_bgmlp.py) -- on CPU in float32 (exact formula, channel order), and on the
GPU through the fused MFMA kernels (bf16 operands, fp32 accumulation: tolerances stated below)."""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT

GOLDEN = os.path.join(ROOT, "tests", "golden", "bgmlp.npz")
# the same module (parameters of bgmlp.npz) on 2 x 64 x 64 pixels = 2 x 16 tiles of the fused kernels, the second image
# with sample coordinates in [-2.5, 2.5]; tests/golden/gen_bgmlp.py
GOLDEN_MULTITILE = os.path.join(ROOT, "tests", "golden", "bgmlp_multitile.npz")
# ... and on ONE ragged image of 29 x 31 = 899 pixels: three full tiles and a fourth with 131 valid rows (round 6)
GOLDEN_RAGGED = os.path.join(ROOT, "tests", "golden", "bgmlp_ragged.npz")


class _Both:
    """Inputs / expected values of one fixture, parameters always from bgmlp.npz."""

    def __init__(self, params, data):
        self.params, self.data = params, data
        self.files = list(params.files) + list(data.files)

    def __getitem__(self, k):
        return self.params[k] if k.startswith("param/") else self.data[k]

    def has(self, k):
        return k in self.data.files


def _load(device, fused=True, fixture=GOLDEN):
    import __graft_entry__  # noqa: F401
    from ava256_amd.trainloop import BackgroundMLPStandIn
    g = _Both(np.load(GOLDEN), np.load(fixture))
    m = BackgroundMLPStandIn(3, 2, fused=fused)
    sd = {}
    for k in g.params.files:
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


@pytest.mark.parametrize("fixture", [GOLDEN, GOLDEN_MULTITILE, GOLDEN_RAGGED])
def test_standin_matches_the_reference_module_on_cpu(fixture):
    m, g = _load("cpu", fixture=fixture)
    bg, grads = _run(m, g, "cpu")
    assert np.abs(bg - g["bg"]).max() <= 1e-4 * np.abs(g["bg"]).max()
    for k, v in grads.items():
        if not g.has("grad/" + k):
            continue
        ref = g["grad/" + k].reshape(v.shape)
        # float32 on both sides, Conv2d (reference) against Linear (stand-in): the summation order differs, and over the
        # 2 M hidden units of the multi-tile fixture a couple of pre-activations within an ulp of zero take the other
        # LeakyReLU slope -- one pixel's whole term in one row of an earlier layer's weight gradient.  ONE such flip is
        # 0.8 / (sqrt(8192) * 16) = 5.5e-4 of a gradient's norm, and that is what is measured here (5-7e-4 on every
        # gradient upstream of the second activation, 1e-6 downstream of it; 3e-3 of the largest element).  Element-wise
        # bound on the small fixture, norm-wise (room for ~10 flips) on the large one.
        if fixture == GOLDEN:
            assert np.abs(v - ref).max() <= 1e-4 * np.abs(ref).max(), k
        assert np.linalg.norm(v - ref) <= 2e-3 * np.linalg.norm(ref), (k, np.linalg.norm(v - ref) / np.linalg.norm(ref))


@pytest.mark.gpu
def test_fused_kernels_match_the_reference_module():
    """bf16 operands and stored activations: the output (mean 100, spread from the 25x scale) is held to 1.2 % of its
    spread.  Gradients: bf16 rounding of a pre-activation near zero flips its LeakyReLU slope (1 <-> 0.2), so against
    the float32 reference ANY bf16 execution of this network sits near 10 % norm-wise in the early layers (eager
    autocast: cosine 0.993 / 12 %, these kernels: 0.996 / 9 %, gpurun_out/r02y); held to cosine >= 0.99 and 15 %, and
    in the test below to no worse than eager bf16 autocast."""
    _check_fused_against(GOLDEN)


@pytest.mark.gpu
def test_fused_kernels_match_the_reference_module_over_many_tiles():
    """The same bounds on the multi-image, multi-tile fixture (2 x 16 tiles of 256 pixels; image 1 has sample
    coordinates up to +-2.5, beyond the +-256-revolution domain of v_sin_f32 at the highest octave)."""
    _check_fused_against(GOLDEN_MULTITILE)


@pytest.mark.gpu
def test_fused_kernels_match_the_reference_module_on_a_ragged_image():
    """The same bounds on the ragged fixture: 899 pixels = 3.5 tiles of the fused kernels, non-zero hidden biases."""
    _check_fused_against(GOLDEN_RAGGED)


def _record(fixture, rows):
    """What the bf16 kernels measure against the reference's fp32 module, per tensor, kept per run (gpurun_out/bgmlp_parity.json
    -> profiles/r0N_bgmlp_parity.json): the bounds below are 'worst measured + margin', and this is where 'measured' is."""
    import json
    path = os.path.join(ROOT, "gpurun_out", "bgmlp_parity.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        doc = json.load(open(path)) if os.path.exists(path) else {}
        doc[os.path.basename(fixture)] = rows
        json.dump(doc, open(path, "w"), indent=1)
    except OSError:
        pass


def _check_fused_against(fixture):
    m, g = _load("cuda", fixture=fixture)
    bg, grads = _run(m, g, "cuda")
    spread = np.abs(g["bg"] - 100.0).max()
    rows = {"output_max_abs_over_spread": float(np.abs(bg - g["bg"]).max() / spread)}
    for k, v in grads.items():
        if g.has("grad/" + k):
            ref = g["grad/" + k].reshape(v.shape)
            rows[k] = {"cosine": _cos(v, ref), "norm_wise": float(np.linalg.norm(v - ref) / np.linalg.norm(ref))}
    _record(fixture, rows)
    # Bounds = the worst value MEASURED on the MI355X for that fixture + margin (profiles/r06_bgmlp_parity.json, written by
    # _record above: output 0.6 / 0.9 / 0.9 % of the spread; worst cosine / norm-wise error of a gradient 0.9930 / 13.0 % on the
    # small fixture -- its smallest tensor, idmod.0.weight --, 0.9945 / 10.8 % over 32 tiles, 0.9968 / 8.0 % on the ragged image).
    # What bf16 operands cost: a pre-activation rounded across zero flips its LeakyReLU slope; eager bf16 autocast of the same
    # module sits at 0.993 / 12 %.
    cmin, nmax = {GOLDEN: (0.992, 0.14), GOLDEN_MULTITILE: (0.9935, 0.12), GOLDEN_RAGGED: (0.9955, 0.095)}[fixture]
    assert np.abs(bg - g["bg"]).max() <= 1.2e-2 * spread, (np.abs(bg - g["bg"]).max(), spread)
    for k, v in grads.items():
        if not g.has("grad/" + k):
            continue
        ref = g["grad/" + k].reshape(v.shape)
        assert _cos(v, ref) >= cmin, (k, _cos(v, ref))
        assert np.linalg.norm(v - ref) <= nmax * np.linalg.norm(ref), (k, np.linalg.norm(v - ref) / np.linalg.norm(ref))


@pytest.mark.gpu
@pytest.mark.parametrize("shape,coord_range", [((2, 96, 80), 1.0), ((1, 37, 53), 1.0), ((3, 128, 128), 1.0),
                                               ((2, 64, 72), 3.5)])
def test_fused_kernels_match_eager_fp32_on_ragged_images(shape, coord_range):
    """Image sizes that are not multiples of the 128-pixel tile; the stand-in's own seeded weights; eager float32 on
    the same device as the reference, and eager bf16 autocast as the yardstick of what the dtype costs: the fused
    kernels may not be further from float32 than 1.25 x the eager bf16 run (+ 1 %).  (The reference of THIS test is
    eager float32 torch on the same device, not a fixture: it checks tiling and raggedness; the arithmetic is pinned by
    the reference-made fixtures above.)  coord_range > 1: sample coordinates outside [-1, 1] (cropped / jittered
    pixelcoords) -- 2^8 x then leaves the +-256-revolution domain of v_sin_f32 / v_cos_f32 and must be range-reduced."""
    import __graft_entry__  # noqa: F401
    from ava256_amd.trainloop import BackgroundMLPStandIn
    B, H, W = shape
    gen = torch.Generator().manual_seed(B * 1000 + H)
    cam, idx = torch.randint(0, 5, (B,), generator=gen).cuda(), torch.randint(0, 3, (B,), generator=gen).cuda()
    sc = ((torch.rand(B, H, W, 2, generator=gen) * 2 - 1) * coord_range).cuda()
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


@pytest.mark.gpu
def test_image_groups_do_not_change_the_result():
    """fused_background_mlp walks a training batch in groups of images (bgmlp.IMAGES_PER_CALL) so that the gradient planes
    exist for one group at a time: same output bits (tiles never span images), weight gradients equal up to the fp32
    summation order over the groups."""
    from ava256_amd import bgmlp
    gen = torch.Generator().manual_seed(5)
    B, H, W = 3, 24, 40
    sc = (torch.rand(B, H, W, 2, generator=gen) * 2 - 1).cuda()
    mk = lambda *shape, s=0.1: (torch.randn(*shape, generator=gen) * s).cuda().requires_grad_(True)
    bias1, w1pos, w6, b6 = mk(B, 256), mk(256, 40), mk(3, 256), mk(3)
    hidden = [(mk(256, 256, s=0.06), mk(256)) for _ in range(4)]
    gout = torch.randn(B, 3, H, W, generator=gen).cuda()
    params = [bias1, w1pos, w6, b6] + [t for wb in hidden for t in wb]

    def run(n):
        for t in params:
            t.grad = None
        out = bgmlp.fused_background_mlp(sc, bias1, w1pos, hidden, w6, b6, images_per_call=n)
        out.backward(gout)
        return out.detach().clone(), [t.grad.detach().clone() for t in params]

    o1, g1 = run(B)   # one call
    o2, g2 = run(1)   # one call per image
    o3, g3 = run(2)   # ragged grouping: 2 + 1
    assert torch.equal(o1, o2) and torch.equal(o1, o3)
    for a, b, c in zip(g1, g2, g3):
        scale = float(a.abs().max()) + 1e-12
        assert float((a - b).abs().max()) <= 2e-3 * scale and float((a - c).abs().max()) <= 2e-3 * scale


def test_grouping_and_train_flag_host_logic(monkeypatch):
    """Host logic of fused_background_mlp (no kernel runs): under torch.no_grad(), or when nothing requires a gradient, ONE call
    with train = False (the inference instantiation, whatever the batch size); in training a batch larger than the group size
    is cut into consecutive image groups, the per-image bias with it, and the pieces are concatenated in order."""
    from ava256_amd import bgmlp
    calls = []

    def fake_apply(train, sc, bias1, w1pos, w6, b6, *flat):
        calls.append((bool(train), sc.shape[0], bias1.shape[0], float(sc[0, 0, 0, 0])))
        return sc[..., 0][:, None].expand(sc.shape[0], 3, sc.shape[1], sc.shape[2]) + bias1.sum() * 0

    monkeypatch.setattr(bgmlp._FusedBgMlp, "apply", staticmethod(fake_apply))
    B = 7
    sc = torch.arange(B, dtype=torch.float32)[:, None, None, None].expand(B, 2, 3, 2).contiguous()
    bias1 = torch.zeros(B, 256, requires_grad=True)
    w1pos, w6, b6 = torch.zeros(256, 40), torch.zeros(3, 256), torch.zeros(3)
    hidden = [(torch.zeros(256, 256), torch.zeros(256)) for _ in range(4)]
    out = bgmlp.fused_background_mlp(sc, bias1, w1pos, hidden, w6, b6, images_per_call=3)
    assert calls == [(True, 3, 3, 0.0), (True, 3, 3, 3.0), (True, 1, 1, 6.0)]
    assert out.shape == (B, 3, 2, 3) and torch.equal(out[:, 0, 0, 0], torch.arange(B, dtype=torch.float32))
    calls.clear()
    with torch.no_grad():
        bgmlp.fused_background_mlp(sc, bias1, w1pos, hidden, w6, b6, images_per_call=3)
    assert calls == [(False, B, B, 0.0)]
    calls.clear()
    bgmlp.fused_background_mlp(sc, bias1.detach(), w1pos, hidden, w6, b6, images_per_call=3)   # nothing requires a gradient
    assert calls == [(False, B, B, 0.0)]
    calls.clear()
    bgmlp.fused_background_mlp(sc[:2], bias1[:2], w1pos, hidden, w6, b6)                        # small batch: one call
    assert calls == [(True, 2, 2, 0.0)]
    calls.clear()
    # pixel coordinates that require grad (and nothing else does): they are data here -- the call sees them detached and runs
    # the inference instantiation; autograd builds no node over a forward that saved nothing (advisor, round 5)
    seen = []
    monkeypatch.setattr(bgmlp._FusedBgMlp, "apply", staticmethod(lambda train, sc_, *a: (seen.append(sc_.requires_grad),
                                                                                        fake_apply(train, sc_, *a))[1]))
    out = bgmlp.fused_background_mlp(sc.clone().requires_grad_(True), bias1.detach(), w1pos, hidden, w6, b6)
    assert calls == [(False, B, B, 0.0)] and seen == [False] and not out.requires_grad
