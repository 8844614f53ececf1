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

MARCH = ["march_k8_m8", "march_k64_m4", "march_k8_m8_sat", "march_warp_k8_m8", "march_warp_k8_m8_sat"]


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
