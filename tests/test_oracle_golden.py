"""Pin the CPU checker (oracle/mvp_oracle.c) against the reference's own dense PyTorch statement.

The fixtures tests/golden/march_*.npz were produced by tests/golden/gen_golden.py, which executes
/root/reference/extensions/mvpraymarch/mvpraymarch.py:553-641 (dense loop + autograd) in float64.
The float64 build of the oracle must reproduce RGBA and every gradient to round-off; the float32
build (same code, kernel arithmetic type) must stay within the fp32 tolerance the GPU tests use.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN

MARCH = ["march_k8_m8", "march_k64_m4", "march_k8_m8_sat", "march_k125_m4_fade", "march_warp_k8_m8", "march_warp_k8_m8_sat"]


def _run(o, g):
    args = (g["raypos"], g["raydir"], float(g["stepsize"]), g["tminmax"], g["primpos"], g["primrot"],
            g["primscale"], g["template"])
    fs, fe = float(g["fadescale"]), float(g["fadeexp"])
    warp = g["warp"] if "warp" in g.files else None   # warp-field sampler goldens (algo 1, mvpraymarch.py:762-774)
    rgba, raysat, st = o.march_forward(*args, fs, fe, warp=warp)
    grads = o.march_backward(*args, raysat, np.ones_like(rgba), fs, fe, warp=warp)
    gp, gr, gs, gt = grads[:4]
    mine = dict(template=gt * g["chain_template"], primpos=gp * g["chain_primpos"], primrot=gr,
                primscale=gs * g["chain_primscale"])
    if warp is not None:
        mine["warp"] = grads[4]
    return rgba, raysat, st, mine


@pytest.mark.parametrize("name", MARCH)
def test_oracle_f64_matches_reference_dense_loop(oracle64, name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    rgba, raysat, st, mine = _run(oracle64, g)
    assert st["list_overflow"] == 0
    assert np.abs(rgba - g["rgba"]).max() <= 1e-11 * max(1.0, np.abs(g["rgba"]).max())
    for k, v in mine.items():
        ref = g["graw_" + k]
        assert np.abs(v - ref).max() <= 1e-10 * np.abs(ref).max(), k
    # raysat rule (primaccum.h:72-77): -1 unless the ray saturated
    sat = rgba[..., 3] >= 1.0
    assert np.all(raysat[~sat] == -1.0)
    assert np.all(raysat[sat][:, 0] > -1.0)


@pytest.mark.parametrize("name", MARCH)
def test_oracle_f32_within_fp32_tolerance(oracle32, name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    rgba, raysat, st, mine = _run(oracle32, g)
    assert np.abs(rgba - g["rgba"]).max() <= 2e-4 * max(1.0, np.abs(g["rgba"]).max())
    ref = g["graw_template"]
    assert np.abs(mine["template"] - ref).max() <= 1e-3 * np.abs(ref).max()
    for k in ("primpos", "primrot", "primscale") + (("warp",) if "warp" in mine else ()):
        ref = g["graw_" + k]
        cos = (mine[k] * ref).sum() / np.sqrt((mine[k] ** 2).sum() * (ref ** 2).sum())
        assert cos >= 0.9999, (k, cos)
        assert np.abs(mine[k] - ref).max() <= 3e-2 * np.abs(ref).max(), k


def test_oracle_raydirs_matches_reference_dense_statement(oracle64):
    g = np.load(os.path.join(GOLDEN, "raydirs_small.npz"))
    # the reference's dense statement has volradius == 1 (extensions/utils/utils.py:91)
    raypos, raydir, tminmax = oracle64.raydirs(g["viewpos"], g["viewrot"], g["focal"], g["princpt"],
                                               g["pixelcoords"], float(g["volradius"]))
    assert np.abs(raypos - g["raypos"]).max() <= 1e-13
    assert np.abs(raydir - g["raydir"]).max() <= 1e-13
    assert np.abs(tminmax - g["tminmax"]).max() <= 1e-11


def test_oracle_aabb_contains_boxes(oracle64):
    rng = np.random.default_rng(0)
    N, K = 2, 13  # non-power-of-two heap
    pos = rng.normal(size=(N, K, 3)) * 0.3
    q, _ = np.linalg.qr(rng.normal(size=(N, K, 3, 3)))
    scale = np.exp(rng.normal(size=(N, K, 3)) * 0.3) * 4
    A = oracle64.aabb(pos, q, scale)
    # every leaf AABB contains its 8 corners; every parent contains its children; root contains all
    for n in range(N):
        for k in range(K):
            c = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], float) / scale[n, k]
            world = c @ q[n, k].T + pos[n, k]  # (dot(p,R0),dot(p,R1),dot(p,R2)) + pos, primtransf.h:16-17
            lo, hi = A[n, K - 1 + k]
            assert np.all(world >= lo - 1e-12) and np.all(world <= hi + 1e-12)
            assert np.allclose(world.min(0), lo) and np.allclose(world.max(0), hi)
        for i in range(K - 1):
            assert np.all(A[n, i, 0] == np.minimum(A[n, 2 * i + 1, 0], A[n, 2 * i + 2, 0]))
            assert np.all(A[n, i, 1] == np.maximum(A[n, 2 * i + 1, 1], A[n, 2 * i + 2, 1]))


def test_boundary_fixture_is_the_oracle_behind_the_reference_glue(oracle64):
    """tests/golden/boundary_raymarcher.npz was produced by the reference's unmodified Python glue around float64
    stand-ins of its native modules (tests/golden/gen_boundary.py).  Re-deriving its outputs from the oracle alone pins
    what the glue adds: stepsize = dt / volradius (mvpraymarcher.py:24), only options that are parameters of
    mvpraymarch pass the renderoptions filter (:45) -- fadescale=6 arrived, the unknown key did not --, and the
    result is the [N,H,W,4] march output permuted to NCHW and split 3 + 1 (:50-51)."""
    g = np.load(os.path.join(GOLDEN, "boundary_raymarcher.npz"))
    calls = list(g["calls"])
    assert calls == ["compute_raydirs_forward", "compute_aabb", "raymarch_forward fadescale=6 fadeexp=8 algo=0 chlast=True",
                     "raymarch_backward"]
    rp, rd, tm = oracle64.raydirs(g["in_campos"], g["in_camrot"], g["in_focal"], g["in_princpt"], g["in_pixelcoords"],
                                  float(g["volradius"]))
    np.testing.assert_allclose(rp, g["raypos"], atol=1e-12)
    np.testing.assert_allclose(rd, g["raydir"], atol=1e-12)
    np.testing.assert_allclose(tm, g["tminmax"], atol=1e-12)
    stepsize = float(g["dt"]) / float(g["volradius"])
    args = (rp, rd, stepsize, tm, g["in_primpos"], g["in_primrot"], g["in_primscale"], g["in_template"])
    rgba, sat, st = oracle64.march_forward(*args, fadescale=6.0, fadeexp=8.0)
    np.testing.assert_allclose(np.transpose(rgba, (0, 3, 1, 2))[:, :3], g["rayrgb"], atol=1e-12)
    np.testing.assert_allclose(np.transpose(rgba, (0, 3, 1, 2))[:, 3:4], g["rayalpha"], atol=1e-12)
    assert 0.3 < st["rays_saturated"] / (rgba.shape[0] * rgba.shape[1] * rgba.shape[2]) < 0.8
    gout = np.concatenate([g["w_rgb"], g["w_a"]], axis=1).transpose(0, 2, 3, 1)  # d loss / d rayrgba, NHWC
    gp, gr, gs, gt = oracle64.march_backward(*args, sat, np.ascontiguousarray(gout), fadescale=6.0, fadeexp=8.0)
    for got, k in ((gp, "primpos"), (gr, "primrot"), (gs, "primscale"), (gt, "template")):
        # (the glue allocates raysat and nodeaabb as float32 whatever the default dtype: mvpraymarch.py:81,146-148)
        np.testing.assert_allclose(got, g["grad_" + k], rtol=1e-5, atol=1e-6 * np.abs(g["grad_" + k]).max())
    # with the default fadescale the image differs: the option really went through the filter
    rgba8, _, _ = oracle64.march_forward(*args, fadescale=8.0, fadeexp=8.0)
    assert np.abs(rgba8 - rgba).max() > 1e-3


def test_oracle_edge_diagnostic(oracle64):
    """The per-ray `edge` diagnostic (mvp_oracle.c, mvpo_set_edge_diagnostics): the opacity increment at stake in the
    sample-inclusion decisions (strict box test, march bound) that came within `edge_eps` world units of flipping.  It
    never changes the render; it is zero when no decision is close (eps 0), grows with eps, is bounded by the largest
    possible increment, and is what an eps-sized displacement of the boxes can change in a ray's opacity."""
    from ava256_amd.scene import make_scene
    s = make_scene(1, 40, 36, 24, device="cpu", seed=4, alpha_gain=6.0, slab=4)
    rp, rd, tm = oracle64.raydirs(s["campos"].numpy(), s["camrot"].numpy(), s["focal"].numpy(), s["princpt"].numpy(),
                                  s["pixelcoords"].numpy(), s["volradius"])
    a = (rp, rd, float(s["stepsize"]), tm, s["primpos"].numpy(), s["primrot"].numpy(), s["primscale"].numpy(),
         s["template"].numpy())
    fade = dict(fadescale=3.0, fadeexp=4.0)  # e^-3 of the opacity is still there AT a box face
    plain, _, _ = oracle64.march_forward(*a, **fade)
    res = {}
    for eps in (0.0, 1e-4, 1e-2):
        rgba, _, st = oracle64.march_forward(*a, ray_diagnostics=True, edge_eps=eps, **fade)
        assert np.array_equal(rgba, plain)
        res[eps] = st["edge"]
    assert res[0.0].max() == 0.0
    assert (res[1e-4] <= res[1e-2]).all() and res[1e-2].max() > 0.0
    amax = np.abs(a[7][..., 3]).max() * a[2]
    assert res[1e-2].max() <= amax * (1 + 1e-12)
    # moving every box by less than eps flips only decisions the diagnostic has flagged: a ray whose alpha changes by
    # more than the smooth variation allows must carry edge > 0 at that eps
    shift = a[4] + 2e-5 * np.array([1.0, -1.0, 0.5])
    moved, _, _ = oracle64.march_forward(a[0], a[1], a[2], a[3], shift, *a[5:], **fade)
    jump = np.abs(moved[..., 3] - plain[..., 3])
    smooth = 2e-3 * np.maximum(plain[..., 3], 1e-3)   # d(alpha)/d(position) * 2e-5 stays far below this
    flagged = res[1e-4] > 0
    assert not (jump > smooth)[~flagged].any(), int((jump > smooth)[~flagged].sum())
