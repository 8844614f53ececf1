"""GPU tests of the corners the round-1 review named: the two forward sweeps must agree bit for bit, the fixed-point
slab-gradient path under heavy-tailed upstream gradients and signed opacity, repeated backward passes over one forward,
and the 512-entry hit-list cap (utils.h:779)."""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT
from helpers import (DEGENERATE_CASES, FragileRays, assert_same_poison, cosine, degenerate_case, npf, scene_rays, to_dev)
from test_gpu_parity import BACKWARD_MODES, FWD_TOL, _check_grads, _march, ops  # noqa: F401  (ops is a fixture)

pytestmark = pytest.mark.gpu

DBG_LIB = os.path.join(ROOT, "build_variants", "libmvp_dbg.so")


def _forward_with_handoff(ops_mod, d, warp=None):
    from ava256_amd import _hooks
    _hooks.keep_raysat = True
    rp, rd, tm = ops_mod.compute_raydirs(d["campos"], d["camrot"], d["focal"], d["princpt"], d["pixelcoords"], d["volradius"])
    t = {k: d[k].clone().requires_grad_(True) for k in ("primpos", "primrot", "primscale", "template")}
    if warp is not None:
        t["warp"] = warp.clone().requires_grad_(True)
    rgba = ops_mod.mvpraymarch(rp, rd, d["stepsize"], tm, (t["primpos"], t["primrot"], t["primscale"]), t["template"],
                               t.get("warp"), algo=1 if warp is not None else 0)
    sat, cnt = _hooks.last_raysat, _hooks.last_pl_count
    _hooks.keep_raysat = False
    _hooks.last_raysat = _hooks.last_pl_count = None
    return rgba, sat, cnt, t


@pytest.mark.skipif(not os.path.exists(DBG_LIB), reason="build_variants/libmvp_dbg.so not built (__graft_entry__.build())")
@pytest.mark.parametrize("cfg", [(2, 128, 128, 512, 1.0), (1, 200, 168, 4096, 20.0), (1, 96, 96, 16384, 6.0), (1, 40, 40, 300, 8.0),
                                 (2, 96, 104, 512, 1.0, (8, 8, 8), 0.05), (1, 72, 64, 300, 8.0, (5, 6, 7), 0.3)],
                         ids=lambda c: "N%d_%dx%d_K%d_a%g" % c[:5] + ("_warp%dx%dx%d_n%g" % (c[5] + (c[6],)) if len(c) > 5 else ""))
def test_lane_independent_sweep_equals_slot_synchronous_sweep(ops, cfg):
    """The forward has two march schedules: the lane-independent sweep (every ray walks its own samples) and the
    slot-synchronous one (the packet steps together; also the fallback beyond the fast path's limits).  Same sample
    set, same order per ray, same arithmetic => rgba, raysat and the per-primitive packet counts must be IDENTICAL.
    The debug build of the library (-DMVP_DEBUG_HOOKS) forces the slot-synchronous sweep through the environment."""
    from ava256_amd import _hooks, _lib
    from ava256_amd.scene import make_scene
    N, H, W, K, again = cfg[:5]
    s = make_scene(N, H, W, K, device="cuda", seed=5 + K, alpha_gain=again)
    warp = None
    if len(cfg) > 5:  # (round 6) the warp-field forward takes both sweeps too: one sampler (march_common.h: sample_warped)
        (WD, WH, WW), noise = cfg[5], cfg[6]
        zz, yy, xx = torch.meshgrid(torch.linspace(-1, 1, WD), torch.linspace(-1, 1, WH), torch.linspace(-1, 1, WW), indexing="ij")
        warp = (torch.stack([xx, yy, zz], dim=-1)[None, None] +
                noise * torch.randn(N, K, WD, WH, WW, 3, generator=torch.Generator().manual_seed(9))).contiguous().cuda()
    diag = torch.zeros(8, dtype=torch.int32, device="cuda")
    _hooks.set_diag_buffer(diag)
    gout = torch.randn(N, H, W, 4, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    rgba1, sat1, cnt1, t1 = _forward_with_handoff(ops, s, warp)
    d1 = _hooks.read_diag()
    assert d1["packets_hit"] > 0 and d1["slowpath_packets"] < d1["packets_hit"], d1   # the fast sweep really ran
    rgba1.backward(gout)
    diag.zero_()
    os.environ["MVP_DEBUG_SLOT_SWEEP"] = "1"
    _lib.use_library(DBG_LIB)
    try:
        rgba2, sat2, cnt2, t2 = _forward_with_handoff(ops, s, warp)
        d2 = _hooks.read_diag()
        rgba2.backward(gout)
    finally:
        del os.environ["MVP_DEBUG_SLOT_SWEEP"]
        _lib.use_library(None)
        _hooks.set_diag_buffer(None)
    assert d2["slowpath_packets"] == d2["packets_hit"] == d1["packets_hit"], (d1, d2)
    assert torch.equal(rgba1, rgba2)
    assert torch.equal(sat1, sat2)
    # The per-primitive record counts: equal -- unless rays saturate.  The lane-independent sweep drops, after its sweep, the
    # records all of whose rays saturated before reaching the primitive (round 6); the slot-synchronous sweep has no per-slot ray
    # masks and keeps them.  Such records name no sample either way.
    c1, c2 = cnt1[: N * K] & 0x3fffffff, cnt2[: N * K] & 0x3fffffff
    assert bool((c1 <= c2).all())
    if again == 1.0:
        assert torch.equal(c1, c2)
    # ... and the backward over either hand-off: same samples; the slab gradient is an integer sum (order-free, so
    # bit-identical although the two forwards append list entries in different orders) whenever both hand-offs name the same
    # packets; where records were dropped, the round's bound is taken over fewer packets and the same sums are rounded on
    # another grid (1e-5 of the largest gradient).  Pose gradients are fp32 sums.
    if torch.equal(c1, c2):
        assert torch.equal(t1["template"].grad, t2["template"].grad)
        if warp is not None:
            assert torch.equal(t1["warp"].grad, t2["warp"].grad)
    else:
        dg = float((t1["template"].grad - t2["template"].grad).abs().max())
        assert dg <= 1e-5 * float(t2["template"].grad.abs().max()), dg
    for k in ("primpos", "primrot", "primscale"):
        g1, g2 = t1[k].grad, t2[k].grad
        assert float((g1 - g2).abs().max()) <= 1e-4 * float(g2.abs().max()), k


@pytest.mark.parametrize("mode", BACKWARD_MODES)
def test_heavy_tailed_upstream_gradient(ops, oracle64, mode):
    """0.1 % of the rays carry an upstream gradient 1e4 times the rest (a few bad pixels in an L1 image loss).  The
    fixed-point scale of a round comes from the ray packets of THAT round's list entries, so primitives the outliers do
    not touch keep full resolution; where an outlier's packet is on the list and the rays marched are far below it, the
    primitive goes to the two-pass (residual) form of the kernel -- NOT to the ray-centric fallback: every primitive's
    slab gradient is held to 2e-4 of ITS OWN max |g| against float64."""
    from ava256_amd.scene import make_scene
    N, H, W, K = 1, 96, 96, 512
    s = make_scene(N, H, W, K, device="cpu", seed=61, alpha_gain=2.0)
    rp, rd, tm = scene_rays(oracle64, s)
    a = (rp, rd, s["stepsize"], tm, s["primpos"].numpy(), s["primrot"].numpy(), s["primscale"].numpy(), s["template"].numpy())
    ref_rgba, ref_sat, st = oracle64.march_forward(*a, ray_diagnostics=True)
    rng = np.random.default_rng(12)
    gout = rng.normal(size=ref_rgba.shape)
    outl = rng.random(size=ref_rgba.shape[:3]) < 1e-3
    assert outl.sum() >= 5
    gout[outl] *= 1.0e4
    fragile = FragileRays(ref_sat, st["margin"], gout, nsamples=st["nsamples"])
    rgba, grads, diag = _march(ops, *a, 8.0, 8.0, grad_out=fragile, mode=mode)
    rgp, rgr, rgs, rgt = oracle64.march_backward(*a, ref_sat, fragile.masked())
    got = grads["template"].reshape(K, -1)
    ref = rgt.reshape(K, -1)
    pmax = np.abs(ref).max(1)
    live = pmax > 0
    rel = np.abs(got - ref).max(1)[live] / pmax[live]
    print("heavy tail (%s): per-primitive relative error max %.2e median %.2e; max|g| spread %.1e" % (
        mode, rel.max(), np.median(rel), pmax[live].max() / pmax[live].min()))
    assert rel.max() <= 2e-4, rel.max()
    assert not (~live).any() or np.abs(got[~live]).max() == 0.0
    for k, refg in (("primpos", rgp), ("primrot", rgr), ("primscale", rgs)):
        assert cosine(grads[k], refg) >= 0.9999, k
    if mode == "prim":  # the precision came from the two-pass kernel, no primitive was pushed to the fp32-atomic fallback
        print("   two-pass primitives %d, handed over %d, flags %#x" % (diag["prims_two_pass"], diag["prims_handed_over"],
                                                                        diag["handoff_flags"]))
        assert diag["prims_two_pass"] > 0 and diag["handoff_flags"] & 8
        assert diag["prims_handed_over"] == 0 and diag["handoff_flags"] & 7 == 0


@pytest.mark.parametrize("mode", BACKWARD_MODES)
def test_signed_opacity(ops, oracle64, mode):
    """The operator accepts any float template (the decoders relu theirs, the API does not): with negative opacity
    the running alpha can go below zero, and the sample that finally saturates a ray gets weight 1 - alpha_before > 1,
    outside the bound the fixed-point accumulators are scaled for.  Such primitives must be detected and come out of
    the ray-centric kernel (fp32 atomics) -- gradients still match the oracle."""
    from ava256_amd.scene import make_scene
    N, H, W, K = 1, 64, 64, 256
    s = make_scene(N, H, W, K, device="cpu", seed=33, alpha_gain=1.0)
    g = torch.Generator().manual_seed(2)
    tpl = s["template"].clone()
    tpl[..., 3] = 120.0 * torch.randn(N, K, 1, 1, 1, generator=g).expand(N, K, 8, 8, 8) + 20.0 * torch.randn(N, K, 8, 8, 8, generator=g)
    rp, rd, tm = scene_rays(oracle64, s)
    a = (rp, rd, s["stepsize"], tm, s["primpos"].numpy(), s["primrot"].numpy(), s["primscale"].numpy(), tpl.numpy())
    ref_rgba, ref_sat, st = oracle64.march_forward(*a, ray_diagnostics=True)
    assert ref_rgba[..., 3].min() < -0.2 and st["rays_saturated"] > 50       # alpha really goes negative, rays saturate
    gout = np.random.default_rng(6).normal(size=ref_rgba.shape)
    fragile = FragileRays(ref_sat, st["margin"], gout, nsamples=st["nsamples"], max_frac=0.01, min_allowed=4)
    rgba, grads, diag = _march(ops, *a, 8.0, 8.0, grad_out=fragile, mode=mode)
    fr = fragile.mask
    err = np.abs(rgba - ref_rgba).max(-1)
    assert (err[~fr] > FWD_TOL * max(1.0, np.abs(ref_rgba).max())).sum() == 0, err[~fr].max()
    rgp, rgr, rgs, rgt = oracle64.march_backward(*a, ref_sat, fragile.masked())
    _check_grads(grads, dict(template=rgt, primpos=rgp, primrot=rgr, primscale=rgs), "signed alpha " + mode)


@pytest.mark.parametrize("mode", BACKWARD_MODES)
def test_warp_field_with_signed_opacity(ops, oracle64, mode):
    """The warp-field variant of the primitive-centric backward has a second set of fixed-point accumulators
    (grad_warp) scaled from the same bounds; a weight outside [-1, 1] (signed opacity) must hand the primitive over
    with BOTH slab gradients zero-filled, and the ray-centric kernel must then produce them (fp32 atomics)."""
    from ava256_amd.scene import make_scene
    N, H, W, K = 1, 64, 64, 256
    s = make_scene(N, H, W, K, device="cpu", seed=34, alpha_gain=1.0)
    g = torch.Generator().manual_seed(3)
    tpl = s["template"].clone()
    tpl[..., 3] = 120.0 * torch.randn(N, K, 1, 1, 1, generator=g).expand(N, K, 8, 8, 8) + 20.0 * torch.randn(N, K, 8, 8, 8, generator=g)
    lin = torch.linspace(-1.0, 1.0, 5)
    zz, yy, xx = torch.meshgrid(lin, lin, lin, indexing="ij")
    warp = (torch.stack([xx, yy, zz], dim=-1)[None, None] + 0.1 * torch.randn(N, K, 5, 5, 5, 3, generator=g)).contiguous().numpy()
    rp, rd, tm = scene_rays(oracle64, s)
    a = (rp, rd, s["stepsize"], tm, s["primpos"].numpy(), s["primrot"].numpy(), s["primscale"].numpy(), tpl.numpy())
    ref_rgba, ref_sat, st = oracle64.march_forward(*a, warp=warp, ray_diagnostics=True)
    assert ref_rgba[..., 3].min() < -0.2 and st["rays_saturated"] > 50
    gout = np.random.default_rng(7).normal(size=ref_rgba.shape)
    fragile = FragileRays(ref_sat, st["margin"], gout, nsamples=st["nsamples"], max_frac=0.01, min_allowed=4)
    rgba, grads, diag = _march(ops, *a, 8.0, 8.0, grad_out=fragile, mode=mode, warp=warp)
    rgp, rgr, rgs, rgt, rgw = oracle64.march_backward(*a, ref_sat, fragile.masked(), warp=warp)
    _check_grads(grads, dict(template=rgt, primpos=rgp, primrot=rgr, primscale=rgs), "signed alpha + warp " + mode)
    gw = grads["warp"]
    assert cosine(gw, rgw) >= 0.9999 and np.linalg.norm(gw - rgw) <= 1e-2 * np.linalg.norm(rgw)


def test_warp_field_backward_is_reproducible(ops):
    """grad_template and grad_warp of the primitive-centric kernel are integer sums: two runs give the same bits."""
    from ava256_amd.scene import make_scene
    N, H, W, K = 2, 96, 96, 512
    s = make_scene(N, H, W, K, device="cuda", seed=12, alpha_gain=3.0)
    g = torch.Generator(device="cuda").manual_seed(4)
    lin = torch.linspace(-1.0, 1.0, 8, device="cuda")
    zz, yy, xx = torch.meshgrid(lin, lin, lin, indexing="ij")
    warp0 = (torch.stack([xx, yy, zz], dim=-1)[None, None] + 0.1 * torch.randn(N, K, 8, 8, 8, 3, device="cuda", generator=g)).contiguous()
    gout = torch.randn(N, H, W, 4, device="cuda", generator=g)
    rp, rd, tm = ops.compute_raydirs(s["campos"], s["camrot"], s["focal"], s["princpt"], s["pixelcoords"], s["volradius"])
    res = []
    for _ in range(2):
        tpl = s["template"].clone().requires_grad_(True)
        w = warp0.clone().requires_grad_(True)
        rgba = ops.mvpraymarch(rp, rd, s["stepsize"], tm, (s["primpos"], s["primrot"], s["primscale"]), tpl, w, algo=1)
        rgba.backward(gout)
        res.append((tpl.grad.clone(), w.grad.clone()))
    assert float(res[0][1].abs().max()) > 0
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


@pytest.mark.parametrize("mode", BACKWARD_MODES)
@pytest.mark.parametrize("N", [5, 8, 10, 17])
def test_block_to_image_mapping_with_eight_or_more_images(ops, oracle64, N, mode):
    """Blocks are mapped to (image, packet) / (image, primitive) in two regimes: the first N - N % 8 images go whole to
    one XCD each (image x, x + 8, ... on XCD x), the other R = N % 8 are shared by F XCDs each (2 for R = 4, 4 for R = 2,
    else 8) with 8 / F of them in flight (csrc/march_common.h: packet_of_block / prim_of_block).  N = 8: whole only; 5: five shared images in five rounds;
    10 (F = 4) and 17 (F = 8): both regimes in one grid; every image, ragged packets included, must match the oracle.
    (The other tests run N = 1..4: F = 8, 4, 8, 2.)"""
    from ava256_amd.scene import make_scene
    H, W, K = 44, 52, 150
    s = make_scene(N, H, W, K, device="cpu", seed=70 + N, alpha_gain=3.0)
    rp, rd, tm = scene_rays(oracle64, s)
    a = (rp, rd, s["stepsize"], tm, s["primpos"].numpy(), s["primrot"].numpy(), s["primscale"].numpy(), s["template"].numpy())
    ref_rgba, ref_sat, st = oracle64.march_forward(*a, ray_diagnostics=True)
    assert st["rays_hit"] > 0
    gout = np.random.default_rng(N).normal(size=ref_rgba.shape)
    fragile = FragileRays(ref_sat, st["margin"], gout, nsamples=st["nsamples"])
    rgba, grads, diag = _march(ops, *a, 8.0, 8.0, grad_out=fragile, mode=mode)
    fr = fragile.mask
    err = np.abs(rgba - ref_rgba).max(-1)
    assert (err[~fr] > FWD_TOL * max(1.0, np.abs(ref_rgba).max())).sum() == 0, err[~fr].max()
    for n in range(N):   # no image may be skipped or rendered twice into another one's slot
        assert np.abs(rgba[n]).max() > 0
    rgp, rgr, rgs, rgt = oracle64.march_backward(*a, ref_sat, fragile.masked())
    _check_grads(grads, dict(template=rgt, primpos=rgp, primrot=rgr, primscale=rgs), "N=%d %s" % (N, mode))


def test_backward_twice_over_one_forward(ops):
    """retain_graph / several losses: the backward marks things in the forward's hand-off buffer (primitives it hands
    to the ray-centric kernel) and derives its fixed-point scales from the upstream gradient of THAT call.  A second
    backward must not inherit either: after a huge-gradient pass, and after a NaN pass, a small clean gradient gives
    exactly what it gives on a fresh forward."""
    from ava256_amd.scene import make_scene
    s = make_scene(2, 64, 64, 256, device="cuda", seed=9, alpha_gain=4.0)
    g = torch.Generator(device="cuda").manual_seed(1)
    small = 1e-3 * torch.randn(2, 64, 64, 4, device="cuda", generator=g)
    huge = 1e6 * torch.randn(2, 64, 64, 4, device="cuda", generator=g)
    poisoned = small.clone()
    poisoned[0, 30, 30, 1] = float("nan")

    rgba, _, cnt, t = _forward_with_handoff(ops, s)
    rgba.backward(small)
    fresh = {k: v.grad.clone() for k, v in t.items()}
    assert all(torch.isfinite(v).all() for v in fresh.values())

    for first in (huge, poisoned):
        rgba, _, cnt, t = _forward_with_handoff(ops, s)
        rgba.backward(first, retain_graph=True)
        for v in t.values():
            v.grad = None
        rgba.backward(small)
        flags = int(cnt[2 * 256].item())
        assert flags == 0, flags                              # nothing stays handed over from the first pass
        assert torch.equal(t["template"].grad, fresh["template"])
        for k in ("primpos", "primrot", "primscale"):  # fp32 sums whose order follows the LDS ticket order: round-off
            assert (t[k].grad - fresh[k]).abs().max().item() <= 1e-4 * fresh[k].abs().max().item(), k


def test_hit_list_cap_matches_the_reference_rule(ops, oracle64):
    """utils.h:779: a warp lists at most 512 primitives, in traversal (DFS leaf) order; later hits are dropped.  Here a
    packet is an 8x8 wave instead of the reference's 8x4 warp; when every ray of the image hits every box, the kept
    set is the same 512 for any packet shape, so the result must equal the oracle with maxhitboxes = 512 -- and the
    drop is counted in diag, never silent."""
    from ava256_amd.scene import make_scene
    N, H, W, K = 1, 16, 16, 700
    s = make_scene(N, H, W, K, device="cpu", seed=3, alpha_gain=1.0)
    g = torch.Generator().manual_seed(4)
    s["primpos"] = (0.01 * torch.randn(N, K, 3, generator=g)).contiguous()
    s["primrot"] = torch.eye(3).expand(N, K, 3, 3).contiguous()
    s["primscale"] = torch.full((N, K, 3), 1.0 / 0.8)
    tpl = s["template"].clone()
    tpl[..., 3] = 0.0002 * (1.0 + torch.rand(N, K, 8, 8, 8, generator=g))     # faint: no ray saturates
    stepsize = 1.0 / 32.0
    rp, rd, tm = scene_rays(oracle64, s)
    a = (rp, rd, stepsize, tm, s["primpos"].numpy(), s["primrot"].numpy(), s["primscale"].numpy(), tpl.numpy())
    ref_rgba, ref_sat, st = oracle64.march_forward(*a, maxhitboxes=512)
    assert st["list_overflow"] == H * W * (K - 512) and st["rays_saturated"] == 0   # every ray hits every box
    gout = np.random.default_rng(2).normal(size=ref_rgba.shape)
    rgba, grads, diag = _march(ops, *a, 8.0, 8.0, grad_out=gout)
    assert diag["list_overflow"] > 0 and diag["max_list"] == 512, diag
    assert np.abs(rgba - ref_rgba).max() <= 4 * FWD_TOL * max(1.0, np.abs(ref_rgba).max())   # 512 x 50 samples per ray
    rgp, rgr, rgs, rgt = oracle64.march_backward(*a, ref_sat, gout, maxhitboxes=512)
    kept = np.abs(rgt).reshape(K, -1).max(1) > 0
    assert kept.sum() == 512
    assert (np.abs(grads["template"]).reshape(K, -1).max(1) > 0).tolist() == kept.tolist()   # the same 512 are kept
    _check_grads(grads, dict(template=rgt, primpos=rgp, primrot=rgr, primscale=rgs), "list cap")


@pytest.mark.parametrize("pixel_form", ["tensor", "tuple"])
def test_fused_camera_entry_point_is_bit_identical(ops, pixel_form):
    """mvp_march_forward_cams makes the rays inside the march.  One shared statement of the ray
    arithmetic (mvp_device.h: ray_from_camera) => the image, raysat, the ray tensors the grad-mode forward writes for
    its backward (every pixel of a ragged image, bit for bit what compute_raydirs writes) and the slab gradient are
    IDENTICAL to compute_raydirs + mvpraymarch (pose gradients to fp32 round-off: their ray sums are not
    order-deterministic in either form).  Both forms of pixelcoords (extensions/utils/utils.py:28-33)."""
    from ava256_amd.scene import make_scene
    N, H, W, K = 3, 83, 101, 512
    s = make_scene(N, H, W, K, device="cuda", seed=17, alpha_gain=8.0)
    pc = s["pixelcoords"] if pixel_form == "tensor" else (W, H)
    g = torch.Generator(device="cuda").manual_seed(2)
    gout = torch.randn(N, H, W, 4, device="cuda", generator=g)
    names = ("primpos", "primrot", "primscale", "template")

    t1 = {k: s[k].clone().requires_grad_(True) for k in names}
    rp, rd, tm = ops.compute_raydirs(s["campos"], s["camrot"], s["focal"], s["princpt"], pc, s["volradius"])
    a = ops.mvpraymarch(rp, rd, s["stepsize"], tm, (t1["primpos"], t1["primrot"], t1["primscale"]), t1["template"], None)
    a.backward(gout)

    t2 = {k: s[k].clone().requires_grad_(True) for k in names}
    b = ops.mvpraymarch_from_cameras(s["campos"], s["camrot"], s["focal"], s["princpt"], pc, s["volradius"], s["stepsize"],
                                     (t2["primpos"], t2["primrot"], t2["primscale"]), t2["template"])
    saved = b.grad_fn.saved_tensors  # (raypos, raydir, tminmax, ...): written by the forward march, read by the backward
    assert torch.equal(saved[0], rp) and torch.equal(saved[1], rd) and torch.equal(saved[2], tm)
    b.backward(gout)
    assert torch.equal(a, b)
    assert torch.equal(t1["template"].grad, t2["template"].grad)
    for k in ("primpos", "primrot", "primscale"):  # fp32 sums over rays in queue order (LDS tickets): equal to round-off
        assert (t1[k].grad - t2[k].grad).abs().max().item() <= 1e-4 * t1[k].grad.abs().max().item(), k
    # the module-level form, no-grad
    rm = ops.Raymarcher(s["volradius"], dt=1.0)
    with torch.no_grad():
        r1 = rm(rp, rd, tm, s)
        r2 = rm.forward_from_cameras(s["campos"], s["camrot"], s["focal"], s["princpt"], pc, s)
    assert torch.equal(r1[0], r2[0]) and torch.equal(r1[1], r2[1])
    with pytest.raises(TypeError):
        ops.mvpraymarch_from_cameras(s["campos"], s["camrot"], s["focal"], s["princpt"], pc, s["volradius"], s["stepsize"],
                                     (s["primpos"], s["primrot"], s["primscale"]), s["template"], not_an_option=1)


def test_list_capacity_follows_the_demand(ops, oracle64):
    """A close-up camera (focal x 4): primitives cover many more ray packets than the first-call heuristic of the list
    capacity allows.  First call: the overflowed primitives go through the ray-centric kernel -- which marches ONLY the
    packets on their lists (per-packet marks), not the whole image; the forward's counters say what would have been
    needed, the operator reads them back without a host synchronisation, and the second call sizes the lists from that:
    no overflow, everything primitive-centric.  Both calls match the float64 oracle."""
    from ava256_amd import _hooks
    import importlib
    op = importlib.import_module("ava256_amd.mvpraymarch")
    from ava256_amd.scene import make_scene
    N, H, W, K = 1, 256, 256, 2048          # C3-like density of primitives per packet, kept small for the oracle
    s = make_scene(N, H, W, K, device="cpu", seed=23, alpha_gain=4.0)
    s["focal"] = s["focal"] * 4.0
    op._LIST_DEMAND.pop((0, H, W, K), None)
    cap0 = op.primlist_capacity(H, W, K, torch.device("cuda", 0))
    rp, rd, tm = scene_rays(oracle64, s)
    a = (rp, rd, s["stepsize"], tm, s["primpos"].numpy(), s["primrot"].numpy(), s["primscale"].numpy(), s["template"].numpy())
    ref_rgba, ref_sat, st = oracle64.march_forward(*a, ray_diagnostics=True)
    gout = np.random.default_rng(4).normal(size=ref_rgba.shape)
    fragile = FragileRays(ref_sat, st["margin"], gout, nsamples=st["nsamples"])
    ref = dict(zip(("primpos", "primrot", "primscale", "template"), oracle64.march_backward(*a, ref_sat, gout)))
    seen = []
    for call in range(2):
        rgba, grads, diag = _march(ops, *a, 8.0, 8.0, grad_out=fragile, mode="prim")
        assert np.abs(rgba - ref_rgba).max() <= FWD_TOL * max(1.0, np.abs(ref_rgba).max())
        ref_m = dict(zip(("primpos", "primrot", "primscale", "template"), oracle64.march_backward(*a, ref_sat, fragile.masked())))
        _check_grads(grads, ref_m, "call %d" % call)
        torch.cuda.synchronize()
        seen.append((diag["handoff_flags"], op.primlist_capacity(H, W, K, torch.device("cuda", 0))))
    (flags1, cap1), (flags2, cap2) = seen
    print("list capacity: heuristic %d -> measured demand -> %d; flags %#x then %#x" % (cap0, cap1, flags1, flags2))
    assert flags1 & 1, "the scene is meant to overflow the heuristic capacity on the first call"
    assert cap1 > cap0 and cap2 == cap1
    assert flags2 & 7 == 0, flags2          # second call: nothing left the primitive-centric path


@pytest.mark.parametrize("mode", BACKWARD_MODES)
@pytest.mark.parametrize("case", DEGENERATE_CASES)
def test_degenerate_primitive_inputs(ops, oracle64, case, mode):
    """Degenerate and non-finite PRIMITIVE inputs (tests/helpers.py: degenerate_case; the reference semantics of every case
    are pinned on CPU in tests/test_oracle_degenerate.py): primscale = 0 (a fresh DecoderAssembler: adaptwarps = 0,
    models/decoders/assembler.py:66,199 -> 1/scale = inf in primtransf.h:12-63), primscale = inf, NaN / Inf in the slab, the
    position or the rotation of ONE primitive (a diverged decoder, ddp-train.py:436-439), rays with tmin = tmax.  The kernels
    must poison exactly the elements the reference's arithmetic poisons (same NaN / Inf pattern in the image and in every
    gradient), be within the standing tolerances everywhere else, in all three backward ownership modes -- and come back
    in bounded time (an all-infinite AABB tree, 500-step crossings)."""
    import time
    c = degenerate_case(case, oracle64)
    a = (c["raypos"], c["raydir"], c["stepsize"], c["tminmax"], c["primpos"], c["primrot"], c["primscale"], c["template"])
    with np.errstate(all="ignore"):
        ref_rgba, ref_sat, st = oracle64.march_forward(*a, ray_diagnostics=True)
    rng = np.random.default_rng(3)
    gout = rng.normal(size=ref_rgba.shape)
    SENT = 12345.0   # NaN-aware comparison of raysat: a NaN on one side only must count as a difference
    margin = np.where(np.isfinite(st["margin"]), st["margin"], 0.0)
    fragile = FragileRays(np.nan_to_num(ref_sat, nan=SENT, posinf=SENT, neginf=-SENT), margin, gout, max_frac=0.01, min_allowed=4)
    t0 = time.time()
    rgba, grads, diag = _march(ops, *a, 8.0, 8.0, mode=mode,
                               grad_out=lambda hs: fragile(np.nan_to_num(hs, nan=SENT, posinf=SENT, neginf=-SENT)))
    elapsed = time.time() - t0
    assert elapsed < 60.0, elapsed
    with np.errstate(all="ignore"):
        gp, gr, gs, gt = oracle64.march_backward(*a, ref_sat, fragile.masked())
    fr = fragile.mask
    assert_same_poison(rgba[~fr], ref_rgba[~fr], 4 * FWD_TOL, case + " rgba")
    nbad = {k: int((~np.isfinite(v)).sum()) for k, v in (("template", gt), ("primpos", gp), ("primrot", gr), ("primscale", gs))}
    print("%s (%s): %.2f s, %d fragile rays, oracle non-finite elements %s, flags %#x" % (
        case, mode, elapsed, int(fr.sum()), nbad, diag.get("handoff_flags", -1)))
    assert_same_poison(grads["template"], gt, GT_TOL_DEGENERATE, case + " grad_template")
    for k, ref in (("primpos", gp), ("primrot", gr), ("primscale", gs)):
        assert_same_poison(grads[k], ref, POSE_TOL_DEGENERATE, case + " grad_" + k)
        fin = np.isfinite(ref) & np.isfinite(grads[k])
        if np.abs(ref[fin]).max() > 0:
            assert cosine(grads[k][fin], ref[fin]) >= 0.9999, (case, k)


GT_TOL_DEGENERATE = 1e-3     # grad_template: the standing 1e-3 of max |g| (finite elements)
POSE_TOL_DEGENERATE = 3e-2   # pose gradients: the standing white-noise bound


@pytest.mark.parametrize("again", [1.0, 40.0, 200.0])
def test_slab_gradient_is_bit_reproducible_whatever_the_append_order(ops, again):
    """The forward appends a primitive's list records in whatever order its packets finish -- another order on every run --, and
    a single-round primitive walks them in that order.  Everything a round's fixed-point scale is made of (its exact sample
    count, the bound over the packets its records NAME) must therefore be independent of which wave looks at which record.
    Round 6 broke that for a day: the bound was taken over the packets of the waves that found live rays, which on a SATURATED
    scene -- where whole records lie behind the rays' saturation points -- depends on the order.  Five renders, all bits equal,
    at opacity x 1 (nothing saturates), x 40 and x 200."""
    from ava256_amd.scene import make_scene
    s = make_scene(3, 256, 256, 1024, device="cuda", seed=77, alpha_gain=again)
    gout = torch.randn(3, 256, 256, 4, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    ref = None
    for _ in range(5):
        rgba, _, _, t = _forward_with_handoff(ops, s)
        rgba.backward(gout)
        g = t["template"].grad
        assert torch.isfinite(g).all() and float(g.abs().max()) > 0
        if ref is None:
            ref = g.clone()
        else:
            assert torch.equal(ref, g)
