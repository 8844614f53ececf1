#!/usr/bin/env python3
"""tools/diag_euler_residual.py [library.so ...] -- the Euler-homogeneity residual of tests/test_gpu_fullsize.py::
test_c2_gradient_properties (sum T_rgb * dL/dT_rgb  vs  sum dL/drgb * rgb, 8 cameras of C2 at opacity x20) for the product
library, for other builds of the same ABI, and for the ray-centric backward (fp32 global atomics): which part of the
residual belongs to the accumulation scheme and which to the forward / backward pair itself."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ava256_amd as ops  # noqa: E402
from ava256_amd import _hooks, _lib  # noqa: E402
from ava256_amd.scene import make_scene  # noqa: E402


def residual(s, gout, sl):
    rp, rd, tm = ops.compute_raydirs(s["campos"][sl], s["camrot"][sl], s["focal"][sl], s["princpt"][sl],
                                     s["pixelcoords"][sl], s["volradius"])
    t = {k: s[k][sl].detach().clone().requires_grad_(True) for k in ("primpos", "primrot", "primscale", "template")}
    rgba = ops.mvpraymarch(rp, rd, s["stepsize"], tm, (t["primpos"], t["primrot"], t["primscale"]), t["template"], None)
    rgba.backward(gout)
    g = t["template"].grad
    terms = s["template"][sl][..., :3].double() * g[..., :3].double()
    lhs, absum = terms.sum().item(), terms.abs().sum().item()
    rhs = (gout[..., :3].double() * rgba[..., :3].double()).sum().item()
    return lhs, rhs, absum, g


if __name__ == "__main__":
    s = make_scene(80, 512, 512, 4096, device="cuda", seed=1112, alpha_gain=20.0)
    sl = slice(0, 8)
    gout = torch.randn(8, 512, 512, 4, device="cuda", generator=torch.Generator(device="cuda").manual_seed(7))
    ref = None
    for lib in [None] + sys.argv[1:]:
        _lib.use_library(lib)
        lhs, rhs, absum, g = residual(s, gout, sl)
        print("%-40s lhs %.4f rhs %.4f diff %+.4f  (%.2e of sum |terms| = %.3e)" % (
            os.path.basename(lib or "product"), lhs, rhs, lhs - rhs, (lhs - rhs) / absum, absum))
        if ref is None:
            ref = g
        else:
            print("   grad_template vs product: max-abs %.3e of max |g| %.3e" % ((g - ref).abs().max().item(), ref.abs().max().item()))
    _lib.use_library(None)
    with _hooks.patched_handoff(ray_centric=True):
        lhs, rhs, absum, g = residual(s, gout[:2], slice(0, 2))
    print("%-40s lhs %.4f rhs %.4f diff %+.4f  (2 cameras)" % ("ray-centric (fp32 atomics)", lhs, rhs, lhs - rhs))
    lhs2, rhs2, absum2, g2 = residual(s, gout[:2], slice(0, 2))
    print("%-40s lhs %.4f rhs %.4f diff %+.4f  (2 cameras)" % ("product, same 2 cameras", lhs2, rhs2, lhs2 - rhs2))
