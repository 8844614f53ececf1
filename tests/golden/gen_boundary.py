#!/usr/bin/env python3
"""Generate tests/golden/boundary_raymarcher.npz: what the REFERENCE'S OWN PYTHON GLUE returns at the drop-in boundary.

Runs ONLY in the build container (needs /root/reference).  The reference's glue -- `compute_raydirs`
(extensions/utils/utils.py:21-51), `mvpraymarch` / `MVPRaymarch` / `build_accel`
(extensions/mvpraymarch/mvpraymarch.py:21-390) and `Raymarcher` (models/raymarchers/mvpraymarcher.py:17-54) -- is
imported unmodified from /root/reference and executed on CPU tensors.  Its two native modules (`utilslib`,
`mvpraymarchlib`: CUDA, not buildable here) are replaced by stand-ins with the SAME positional signatures
(mvpraymarch.cpp:146-396, utils.cpp:46-82) whose arithmetic is this repo's float64 oracle (oracle/mvp_oracle.c, itself
pinned to the reference's dense statement).  So the fixture pins everything the glue adds around the kernels:
argument order, the fixed-order tree tensors, dt / volradius, the `co_varnames` filter on renderoptions, the permute /
channel split of the result, which tensors receive gradients, and their values.

The GPU test (tests/test_gpu_parity.py::test_drop_in_boundary_matches_reference_glue) runs this repo's operators
through the same import paths on the same inputs and compares.

Usage:  python tests/golden/gen_boundary.py
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle.mvp_oracle import Oracle  # noqa: E402
from ava256_amd.scene import make_scene  # noqa: E402  (seeded inputs only; no kernels involved)

orc = Oracle("f64")
calls = []


def _np(t):
    return None if t is None else t.detach().cpu().numpy().astype(np.float64)


def _put(dst, arr):
    dst.copy_(torch.from_numpy(np.asarray(arr)).to(dst.dtype))


# ---- utilslib stand-in: utils.cpp:46-82 ---------------------------------------------------------------------------
def compute_raydirs_forward(viewpos, viewrot, focal, princpt, pixelcoords, W, H, volradius, raypos, raydir, tminmax):
    calls.append("compute_raydirs_forward")
    rp, rd, tm = orc.raydirs(_np(viewpos), _np(viewrot), _np(focal), _np(princpt), _np(pixelcoords), float(volradius),
                             hw=(H, W))
    _put(raypos, rp), _put(raydir, rd), _put(tminmax, tm)


def compute_raydirs_backward(*a):
    calls.append("compute_raydirs_backward")


# ---- mvpraymarchlib stand-in: mvpraymarch.cpp:146-396 ---------------------------------------------------------------
def compute_aabb(primpos, primrot, primscale, sortedobjid, nodechildren, nodeparent, nodeaabb, algo):
    calls.append("compute_aabb")
    N, K = primpos.shape[:2]
    assert sortedobjid.dtype == torch.int32 and tuple(sortedobjid.shape) == (N, K)
    assert torch.equal(sortedobjid, torch.arange(K, dtype=torch.int32)[None].expand(N, K))  # fixed order
    assert tuple(nodechildren.shape) == (N, 2 * K - 1, 2) and tuple(nodeparent.shape) == (N, 2 * K - 1)
    _put(nodeaabb, orc.aabb(_np(primpos), _np(primrot), _np(primscale)))


def raymarch_forward(raypos, raydir, stepsize, tminmax, sortedobjid, nodechildren, nodeaabb, primpos, primrot,
                     primscale, template, warp, rayrgba, raysat, rayterm, algo, sortprims, maxhitboxes, synchitboxes,
                     chlast, fadescale, fadeexp, accum, termthresh, griddim, bsx, bsy):
    calls.append("raymarch_forward fadescale=%g fadeexp=%g algo=%d chlast=%s" % (fadescale, fadeexp, algo, chlast))
    assert chlast and accum == 0 and rayterm is None
    rgba, sat, _ = orc.march_forward(_np(raypos), _np(raydir), float(stepsize), _np(tminmax), _np(primpos), _np(primrot),
                                     _np(primscale), _np(template), fadescale=fadescale, fadeexp=fadeexp,
                                     maxhitboxes=maxhitboxes, want_raysat=raysat is not None,
                                     nodeaabb=_np(nodeaabb), warp=_np(warp) if algo == 1 else None)
    _put(rayrgba, rgba)
    if raysat is not None:
        _put(raysat, sat)


def raymarch_backward(raypos, raydir, stepsize, tminmax, sortedobjid, nodechildren, nodeaabb, primpos, grad_primpos,
                      primrot, grad_primrot, primscale, grad_primscale, template, grad_template, warp, grad_warp,
                      rayrgba, grad_rayrgba, raysat, rayterm, algo, sortprims, maxhitboxes, synchitboxes, chlast,
                      fadescale, fadeexp, accum, termthresh, griddim, bsx, bsy):
    calls.append("raymarch_backward")
    g = orc.march_backward(_np(raypos), _np(raydir), float(stepsize), _np(tminmax), _np(primpos), _np(primrot),
                           _np(primscale), _np(template), _np(raysat), _np(grad_rayrgba), fadescale=fadescale,
                           fadeexp=fadeexp, maxhitboxes=maxhitboxes, nodeaabb=_np(nodeaabb),
                           warp=_np(warp) if algo == 1 else None)
    for dst, src in zip((grad_primpos, grad_primrot, grad_primscale, grad_template), g[:4]):
        dst.add_(torch.from_numpy(src).to(dst.dtype))  # the reference kernels accumulate into zero-filled buffers


def main():
    torch.set_default_dtype(torch.float64)
    ul, ml = types.ModuleType("utilslib"), types.ModuleType("mvpraymarchlib")
    ul.compute_raydirs_forward, ul.compute_raydirs_backward = compute_raydirs_forward, compute_raydirs_backward
    ml.compute_aabb, ml.raymarch_forward, ml.raymarch_backward = compute_aabb, raymarch_forward, raymarch_backward
    sys.modules["utilslib"], sys.modules["mvpraymarchlib"] = ul, ml
    # the reference tree wins over this repo's same-named shim packages (extensions/, models/)
    sys.path[:] = [REF] + [p for p in sys.path if os.path.abspath(p or ".") != ROOT]
    for m in [m for m in sys.modules if m.split(".")[0] in ("extensions", "models")]:
        del sys.modules[m]
    from extensions.utils.utils import compute_raydirs
    from models.raymarchers.mvpraymarcher import Raymarcher
    import extensions.mvpraymarch.mvpraymarch as refm
    assert refm.__file__.startswith(REF), refm.__file__

    N, H, W, K = 2, 24, 20, 8
    s = make_scene(N, H, W, K, device="cpu", seed=2024, alpha_gain=0.3, slab=8)
    s["primscale"] = s["primscale"] * 0.5
    dd = lambda t: t.double().contiguous()
    cam = {k: dd(s[k]) for k in ("campos", "camrot", "focal", "princpt", "pixelcoords")}
    volradius = float(s["volradius"])
    raypos, raydir, tminmax = compute_raydirs(cam["campos"], cam["camrot"], cam["focal"], cam["princpt"],
                                              cam["pixelcoords"], volradius)
    decout = {k: dd(s[k]).requires_grad_(True) for k in ("primpos", "primrot", "primscale", "template")}
    renderoptions = {"fadescale": 6.0, "fadeexp": 8.0, "not_an_option_of_mvpraymarch": 123}  # mvpraymarcher.py:45
    rm = Raymarcher(volradius, dt=1.0)
    rayrgb, rayalpha, rayrgba_view, pos_img = rm(raypos, raydir, tminmax, decout, renderoptions=renderoptions)
    assert pos_img is None
    rng = np.random.default_rng(12)
    w_rgb = torch.from_numpy(rng.normal(size=tuple(rayrgb.shape)))
    w_a = torch.from_numpy(rng.normal(size=tuple(rayalpha.shape)))
    loss = (rayrgb * w_rgb).sum() + (rayalpha * w_a).sum()
    loss.backward()
    out = dict(N=N, H=H, W=W, K=K, volradius=volradius, dt=1.0,
               renderoptions_keys=np.array(sorted(renderoptions)), fadescale=6.0, fadeexp=8.0,
               raypos=_np(raypos), raydir=_np(raydir), tminmax=_np(tminmax), rayrgb=_np(rayrgb), rayalpha=_np(rayalpha),
               rayrgba_view_shape=np.array(rayrgba_view.shape), w_rgb=_np(w_rgb), w_a=_np(w_a), loss=float(loss),
               calls=np.array(calls))
    for k, v in cam.items():
        out["in_" + k] = _np(v)
    for k, v in decout.items():
        out["in_" + k] = _np(v)
        out["grad_" + k] = _np(v.grad)
    frac_sat = float((out["rayalpha"] >= 1.0 - 1e-9).mean())
    np.savez_compressed(os.path.join(HERE, "boundary_raymarcher.npz"), **out)
    print("calls:", calls)
    print("rayrgb", out["rayrgb"].shape, "alpha max", out["rayalpha"].max(), "saturated frac", frac_sat,
          "|grad_template|", np.abs(out["grad_template"]).max())


if __name__ == "__main__":
    main()
