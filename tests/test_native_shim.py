"""The native-module shims (ava-256_amd/native_shim.py, extensions/*/mvpraymarchlib.py, utilslib.py): positional
signatures equal to the reference's pybind functions (checked against the mounted .cpp in the build container), and -- on
the GPU -- the call sequence of the reference's autograd Function reproduces this build's operator path."""
import inspect
import os
import re

import numpy as np
import pytest
import torch

REF = "/root/reference"

_RENAME = {"rayposim": "raypos", "raydirim": "raydir", "tminmaxim": "tminmax", "tplate": "template",
           "grad_tplate": "grad_template", "rayrgbaim": "rayrgba", "raysatim": "raysat", "raytermim": "rayterm",
           "viewposim": "viewpos", "viewrotim": "viewrot", "focalim": "focal", "princptim": "princpt",
           "pixelcoordsim": "pixelcoords", "algorithm": "algo", "sortboxes": "sortprims"}


def _cpp_params(path, fn):
    """Ordered parameter names of `fn(...)` as declared in the reference's binding source."""
    src = open(path).read()
    m = re.search(r"\b%s\s*\(([^{;]*?)\)\s*\{" % fn, src, re.S)
    assert m, fn
    names = []
    for part in m.group(1).replace("\n", " ").split(","):
        part = part.split("=")[0].strip()
        if part:
            names.append(_RENAME.get(part.split()[-1].lstrip("*&"), part.split()[-1].lstrip("*&")))
    return names


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference is mounted only in the build container")
@pytest.mark.parametrize("fn,path", [("compute_aabb", "extensions/mvpraymarch/mvpraymarch.cpp"),
                                     ("raymarch_forward", "extensions/mvpraymarch/mvpraymarch.cpp"),
                                     ("raymarch_backward", "extensions/mvpraymarch/mvpraymarch.cpp"),
                                     ("compute_raydirs_forward", "extensions/utils/utils.cpp")])
def test_shim_signatures_equal_the_reference_bindings(fn, path):
    from ava256_amd import native_shim
    ref = _cpp_params(os.path.join(REF, path), fn)
    mine = list(inspect.signature(getattr(native_shim, fn)).parameters)
    mine = ["raydir" if n == "raydirs" else n for n in mine]
    assert mine == ref, (fn, mine, ref)


def test_shim_modules_export_the_reference_names():
    import extensions.mvpraymarch.mvpraymarchlib as m
    import extensions.utils.utilslib as u
    for n in ("compute_morton", "build_tree", "compute_aabb", "raymarch_forward", "raymarch_backward"):   # mvpraymarch.cpp:398-405
        assert callable(getattr(m, n))
    for n in ("compute_raydirs_forward", "compute_raydirs_backward"):                                      # utils.cpp:134-137
        assert callable(getattr(u, n))
    with pytest.raises(NotImplementedError):
        m.compute_morton(None, None, 0)


@pytest.mark.gpu
def test_reference_call_sequence_through_the_shims():
    """The statements of the reference's ComputeRaydirs / build_accel / MVPRaymarch around their native calls
    (extensions/utils/utils.py:24-42, mvpraymarch.py:44-84,141-200,240-282), with the shims as the native modules."""
    import ava256_amd as ops
    import extensions.mvpraymarch.mvpraymarchlib as mvpraymarchlib
    import extensions.utils.utilslib as utilslib
    from ava256_amd.scene import make_scene
    N, H, W, K = 2, 72, 88, 256
    s = make_scene(N, H, W, K, device="cuda", seed=5, alpha_gain=6.0)
    dev = s["primpos"].device
    # --- compute_raydirs (utils.py:24-42) ---
    raypos = torch.empty((N, H, W, 3), device=dev)
    raydir = torch.empty((N, H, W, 3), device=dev)
    tminmax = torch.empty((N, H, W, 2), device=dev)
    utilslib.compute_raydirs_forward(s["campos"], s["camrot"], s["focal"], s["princpt"], s["pixelcoords"], W, H,
                                     s["volradius"], raypos, raydir, tminmax)
    rp, rd, tm = ops.compute_raydirs(s["campos"], s["camrot"], s["focal"], s["princpt"], s["pixelcoords"], s["volradius"])
    assert torch.equal(raypos, rp) and torch.equal(raydir, rd) and torch.equal(tminmax, tm)
    # --- build_accel, fixedorder (mvpraymarch.py:44-84) ---
    primtransfin = (s["primpos"], s["primrot"], s["primscale"])
    sortedobjid = (torch.arange(N * K, dtype=torch.int32, device=dev) % K).view(N, K)
    nodechildren = torch.zeros((N, K + K - 1, 2), dtype=torch.int32, device=dev)   # contents unused (implicit heap)
    nodeparent = torch.zeros((N, K + K - 1), dtype=torch.int32, device=dev)
    nodeaabb = torch.empty((N, K + K - 1, 2, 3), dtype=torch.float32, device=dev)
    mvpraymarchlib.compute_aabb(*primtransfin, sortedobjid, nodechildren, nodeparent, nodeaabb, 0)
    # --- MVPRaymarch.forward (mvpraymarch.py:141-200) ---
    rayrgba = torch.empty((N, H, W, 4), device=dev)
    raysat = torch.full((N, H, W, 3), -1, dtype=torch.float32, device=dev)
    opt = dict(algo=0, sortprims=False, maxhitboxes=512, synchitboxes=True, chlast=True, fadescale=8.0, fadeexp=8.0,
               accum=0, termthresh=0.0, griddim=3)
    mvpraymarchlib.raymarch_forward(raypos, raydir, s["stepsize"], tminmax, sortedobjid, nodechildren, nodeaabb,
                                    *primtransfin, s["template"], None, rayrgba, raysat, None, opt["algo"],
                                    opt["sortprims"], opt["maxhitboxes"], opt["synchitboxes"], opt["chlast"],
                                    opt["fadescale"], opt["fadeexp"], opt["accum"], opt["termthresh"], opt["griddim"], 8, 16)
    # --- MVPRaymarch.backward (mvpraymarch.py:240-282) ---
    g = torch.Generator(device="cuda").manual_seed(3)
    grad_rayrgba = torch.randn(N, H, W, 4, device=dev, generator=g)
    grad_primpos, grad_primrot = torch.zeros_like(s["primpos"]), torch.zeros_like(s["primrot"])
    grad_primscale, grad_template = torch.zeros_like(s["primscale"]), torch.zeros_like(s["template"])
    mvpraymarchlib.raymarch_backward(raypos, raydir, s["stepsize"], tminmax, sortedobjid, nodechildren, nodeaabb,
                                     s["primpos"], grad_primpos, s["primrot"], grad_primrot, s["primscale"],
                                     grad_primscale, s["template"], grad_template, None, None, rayrgba,
                                     grad_rayrgba.contiguous(), raysat, None, opt["algo"], opt["sortprims"],
                                     opt["maxhitboxes"], opt["synchitboxes"], opt["chlast"], opt["fadescale"],
                                     opt["fadeexp"], opt["accum"], opt["termthresh"], opt["griddim"], 8, 16)
    # --- the operator path of this build on the same inputs ---
    t = {k: s[k].clone().requires_grad_(True) for k in ("primpos", "primrot", "primscale", "template")}
    ref = ops.mvpraymarch(rp, rd, s["stepsize"], tm, (t["primpos"], t["primrot"], t["primscale"]), t["template"], None)
    ref.backward(grad_rayrgba)
    assert torch.equal(rayrgba, ref.detach())
    assert torch.equal(grad_template, t["template"].grad)      # primitive-centric path taken by the shim as well
    for mine, k in ((grad_primpos, "primpos"), (grad_primrot, "primrot"), (grad_primscale, "primscale")):
        assert (mine - t[k].grad).abs().max().item() <= 1e-4 * t[k].grad.abs().max().item(), k
