"""The opt-in render path over half-precision slabs (ava-256_amd/halfslab.py, csrc/march_common.h: sample_slab_h).

Parity is exact in kind: the kernel over fp16 slabs is compared with the float64 oracle run ON THE
SAME ROUNDED SLABS (the fp16 tensor read back and widened -- exact) within the standing forward tolerance
  FWD_TOL      max-abs err <= 2e-4 * max(1, max|rgba|)        (tests/test_gpu_parity.py)
and, separately, with the unrounded fp32 render within a stated STORAGE tolerance
  STORAGE_TOL  max-abs err <= 1.5e-3 * max(1, max|rgba|)      (fp16: 2^-11 = 4.9e-4 relative per voxel value; a ray adds
                                                              ~40 samples whose opacity errors also move the weights)
The headline metric and every training path keep fp32 slabs."""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT
from helpers import npf, scene_rays

pytestmark = pytest.mark.gpu

FWD_TOL = 2e-4
STORAGE_TOL = 1.5e-3
DBG_LIB = os.path.join(ROOT, "build_variants", "libmvp_dbg.so")


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import ava256_amd
    return ava256_amd


def _scene(N, H, W, K, again, seed=1112):
    from ava256_amd.scene import make_scene
    return make_scene(N, H, W, K, device="cpu", seed=seed, alpha_gain=again)


def test_template_to_half_rounds_to_nearest_even(ops):
    from ava256_amd import halfslab
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.normal(size=4096) * 200.0, [0.0, -0.0, 65504.0, 70000.0, 1e-8, 2049.0, 2051.0, np.inf]]).astype(np.float32)
    x = np.resize(x, (1, 2, 8, 8, 8, 4)).astype(np.float32)
    t = torch.from_numpy(x).cuda()
    h = halfslab.template_to_half(t)
    assert h.dtype == torch.float16 and h.shape == t.shape
    assert torch.equal(h, t.to(torch.float16))       # torch's cast is round-to-nearest-even with overflow to inf


def test_assemble_half_equals_assemble_then_round(ops):
    from ava256_amd import halfslab
    from ava256_amd.assemble import assemble_template
    g = torch.Generator(device="cuda").manual_seed(5)
    nb, B = 16, 8
    tex = torch.randn(2, 3 * B, 4 * B, 4 * B, device="cuda", generator=g) * 3.0
    op = torch.randn(2, B, 4 * B, 4 * B, device="cuda", generator=g)
    full = assemble_template(tex, op, nb, B)
    half = halfslab.assemble_template_half(tex, op, nb, B)
    assert torch.equal(half, full.to(torch.float16))
    assert torch.equal(half, halfslab.template_to_half(full))


CONFIGS = [
    # name, N, H, W, K, alpha gain
    ("small_unsat", 2, 64, 64, 512, 1.0),
    ("ragged_half_saturated", 1, 50, 37, 512, 40.0),
    ("K_not_pow2", 2, 40, 40, 37, 8.0),
    ("C1", 4, 128, 128, 512, 1.0),
    ("C2_camera", 1, 512, 512, 4096, 1.0),
    ("C2_camera_x20", 1, 512, 512, 4096, 20.0),
    ("C3_camera", 1, 512, 512, 16384, 1.0),
]


@pytest.mark.parametrize("cfg", CONFIGS, ids=lambda c: c[0])
@pytest.mark.parametrize("rays", ["tensors", "cameras"])
def test_half_slab_render_against_oracle_on_the_rounded_slabs(ops, oracle64, cfg, rays):
    from ava256_amd import _hooks, halfslab
    name, N, H, W, K, again = cfg
    s = _scene(N, H, W, K, again)
    if K < 100:
        s["primscale"] = s["primscale"] * 0.5
    d = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in s.items()}
    th = halfslab.template_to_half(d["template"])
    rounded = th.float().cpu().numpy()                          # what the kernel reads, widened exactly
    rp, rd, tm = scene_rays(oracle64, s)
    a = (rp, rd, s["stepsize"], tm, s["primpos"].numpy(), s["primrot"].numpy(), s["primscale"].numpy())
    ref, ref_sat, st = oracle64.march_forward(*a, rounded, ray_diagnostics=True)
    assert st["rays_hit"] > 0 and st["list_overflow"] == 0
    diag = torch.zeros(8, dtype=torch.int32, device="cuda")
    _hooks.set_diag_buffer(diag)
    try:
        with torch.no_grad():
            prim = (d["primpos"], d["primrot"], d["primscale"])
            if rays == "tensors":
                rp_d, rd_d, tm_d = ops.compute_raydirs(d["campos"], d["camrot"], d["focal"], d["princpt"], d["pixelcoords"],
                                                      d["volradius"])
                out = halfslab.render_half(rp_d, rd_d, d["stepsize"], tm_d, prim, th)
                full = ops.mvpraymarch(rp_d, rd_d, d["stepsize"], tm_d, prim, d["template"], None)
            else:
                out = halfslab.render_half_from_cameras(d["campos"], d["camrot"], d["focal"], d["princpt"], d["pixelcoords"],
                                                        d["volradius"], d["stepsize"], prim, th)
                full = ops.mvpraymarch_from_cameras(d["campos"], d["camrot"], d["focal"], d["princpt"], d["pixelcoords"],
                                                    d["volradius"], d["stepsize"], prim, d["template"])
        torch.cuda.synchronize()
        dg = _hooks.read_diag()
    finally:
        _hooks.set_diag_buffer(None)
    assert dg["packets_hit"] > 0 and dg["list_overflow"] == 0
    out, full = npf(out), npf(full)
    scale = max(1.0, np.abs(ref).max())
    # rays the ORACLE calls borderline (alpha passes 1.0 by less than fp32 round-off) may saturate one sample apart
    ok = st["margin"] >= 1e-4
    err = np.abs(out - ref).max(-1)
    assert (err[ok] > FWD_TOL * scale).sum() == 0, (name, float(err[ok].max()), scale)
    assert (~ok).sum() <= max(2, 0.03 * (st["nsamples"] > 0).sum())   # (the oracle's own margin: up to a few % of the hitting rays of an opaque scene)
    assert np.abs(out[..., 3] - ref[..., 3]).max() <= FWD_TOL
    # storage tolerance against the unrounded fp32 render of the same scene (same schedule, fp32 slabs)
    serr = np.abs(out - full).max(-1)
    frac_bad = (serr > STORAGE_TOL * scale).mean()
    print(name, rays, "oracle-on-rounded err %.2e, storage err max %.2e (tol %.2e), rays beyond: %.2e" % (
        err[ok].max() / scale, serr.max() / scale, STORAGE_TOL, frac_bad))
    assert frac_bad <= 1e-3, (name, float(serr.max()), scale)   # (a ray saturating one sample apart jumps by that sample)


@pytest.mark.skipif(not os.path.exists(DBG_LIB), reason="build_variants/libmvp_dbg.so not built (__graft_entry__.build())")
def test_half_slab_sweeps_agree_bit_for_bit(ops):
    """Both forward schedules call the one fp16 sampler: a ray's value does not depend on the packet it sits in."""
    from ava256_amd import _hooks, _lib, halfslab
    s = _scene(1, 200, 168, 4096, 20.0, seed=4101)
    d = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in s.items()}
    th = halfslab.template_to_half(d["template"])
    prim = (d["primpos"], d["primrot"], d["primscale"])
    args = (d["campos"], d["camrot"], d["focal"], d["princpt"], d["pixelcoords"], d["volradius"], d["stepsize"], prim, th)
    diag = torch.zeros(8, dtype=torch.int32, device="cuda")
    _hooks.set_diag_buffer(diag)
    with torch.no_grad():
        a = halfslab.render_half_from_cameras(*args)
    d1 = _hooks.read_diag()
    diag.zero_()
    os.environ["MVP_DEBUG_SLOT_SWEEP"] = "1"
    _lib.use_library(DBG_LIB)
    try:
        with torch.no_grad():
            b = halfslab.render_half_from_cameras(*args)
        d2 = _hooks.read_diag()
    finally:
        del os.environ["MVP_DEBUG_SLOT_SWEEP"]
        _lib.use_library(None)
        _hooks.set_diag_buffer(None)
    assert d1["slowpath_packets"] < d1["packets_hit"] == d2["packets_hit"] == d2["slowpath_packets"], (d1, d2)
    assert torch.equal(a, b)


def test_half_slab_tile_independence_and_full_c2_batch(ops):
    """All 80 cameras of C2 in one launch (the bench's own): finite, and images 0 and 79 are bit for bit the camera
    rendered alone (block -> image mapping of the whole-image regime)."""
    from ava256_amd import halfslab
    from ava256_amd.scene import make_scene
    s = make_scene(80, 512, 512, 4096, device="cuda", seed=1112)
    th = halfslab.template_to_half(s["template"])
    prim = (s["primpos"], s["primrot"], s["primscale"])

    def render(sl):
        with torch.no_grad():
            return halfslab.render_half_from_cameras(s["campos"][sl], s["camrot"][sl], s["focal"][sl], s["princpt"][sl],
                                                    s["pixelcoords"][sl], s["volradius"], s["stepsize"],
                                                    tuple(p[sl].contiguous() for p in prim), th[sl].contiguous())
    full = render(slice(0, 80))
    assert bool(torch.isfinite(full).all())
    assert float(full[..., 3].max()) > 0.05
    for i in (0, 79):
        assert torch.equal(full[i:i + 1], render(slice(i, i + 1)))


def test_half_path_refuses_gradients_and_other_slab_sizes(ops):
    from ava256_amd import halfslab
    s = _scene(1, 16, 16, 8, 1.0)
    d = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in s.items()}
    th = halfslab.template_to_half(d["template"])
    prim = (d["primpos"].clone().requires_grad_(True), d["primrot"], d["primscale"])
    args = (d["campos"], d["camrot"], d["focal"], d["princpt"], d["pixelcoords"], d["volradius"], d["stepsize"])
    with pytest.raises(RuntimeError, match="renders only"):
        halfslab.render_half_from_cameras(*args, prim, th)
    with torch.no_grad():
        halfslab.render_half_from_cameras(*args, prim, th)
        with pytest.raises(RuntimeError, match="float16"):
            halfslab.render_half_from_cameras(*args, prim, d["template"])
        with pytest.raises(NotImplementedError):
            halfslab.render_half_from_cameras(*args, prim, torch.zeros(1, 8, 4, 4, 4, 4, device="cuda", dtype=torch.float16))


def test_half_path_checks_shapes_like_the_fp32_operator(ops):
    """One avatar ([1,K,..] primitives) rendered from N > 1 cameras must be refused, as mvpraymarch refuses it
    (mvpraymarch.py:112-127): the node boxes are sized from primpos.size(0) and the kernel would read past them."""
    from ava256_amd import halfslab
    s = _scene(2, 16, 16, 8, 1.0)
    d = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in s.items()}
    th = halfslab.template_to_half(d["template"])
    cam = (d["campos"], d["camrot"], d["focal"], d["princpt"], d["pixelcoords"], d["volradius"], d["stepsize"])
    one = (d["primpos"][:1], d["primrot"][:1], d["primscale"][:1])
    with torch.no_grad():
        halfslab.render_half_from_cameras(*cam, (d["primpos"], d["primrot"], d["primscale"]), th)
        with pytest.raises(AssertionError):
            halfslab.render_half_from_cameras(*cam, one, th)
        with pytest.raises(AssertionError):
            halfslab.render_half_from_cameras(*cam, one, th[:1])
        with pytest.raises(AssertionError):   # rotation given as [N,K,9]
            halfslab.render_half_from_cameras(*cam, (d["primpos"], d["primrot"].reshape(2, -1, 9), d["primscale"]), th)
        with pytest.raises(AssertionError):   # pixel coordinates of another batch size
            halfslab.render_half_from_cameras(d["campos"], d["camrot"], d["focal"], d["princpt"], d["pixelcoords"][:1],
                                              d["volradius"], d["stepsize"], (d["primpos"], d["primrot"], d["primscale"]), th)
        rp, rd, tm = ops.compute_raydirs(d["campos"], d["camrot"], d["focal"], d["princpt"], d["pixelcoords"], d["volradius"])
        halfslab.render_half(rp, rd, d["stepsize"], tm, (d["primpos"], d["primrot"], d["primscale"]), th)
        with pytest.raises(AssertionError):
            halfslab.render_half(rp, rd, d["stepsize"], tm, one, th[:1])
        with pytest.raises(AssertionError):
            halfslab.render_half(rp.reshape(2, -1, 3), rd.reshape(2, -1, 3), d["stepsize"], tm.reshape(2, -1, 2),
                                 (d["primpos"], d["primrot"], d["primscale"]), th)
