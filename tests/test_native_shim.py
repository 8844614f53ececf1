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


def test_handoff_buffers_are_bounded_in_bytes():
    """native_shim keeps the forward's hand-off buffers keyed by the rayrgba storage; a caller that keeps grad-mode images
    alive must not keep every forward's buffers with them: over HANDOFF_BYTES_MAX the oldest entries go (their backward
    then takes the ray-centric kernel), the newest always stays, and a dying storage removes its own entry."""
    from ava256_amd import native_shim as ns
    old_max, old = ns.HANDOFF_BYTES_MAX, dict(ns._HANDOFF)
    ns._HANDOFF.clear()
    try:
        ns.HANDOFF_BYTES_MAX = 3000
        imgs = [torch.zeros(4, 4) for _ in range(5)]
        bufs = lambda: (torch.zeros(100), torch.zeros(50), torch.zeros(100), 8)     # 1000 bytes
        for im in imgs:
            ns._handoff_put(im, (1, 2, 2, 3), bufs())
        assert len(ns._HANDOFF) == 3                                   # 5000 bytes offered, 3000 kept: the oldest two went
        assert ns._handoff_take(imgs[0], (1, 2, 2, 3))[0] is None      # evicted -> ray-centric backward
        assert ns._handoff_take(imgs[4], (1, 2, 2, 3))[3] == 8         # the newest is there
        assert ns._handoff_take(imgs[4], (1, 2, 2, 4))[0] is None      # other geometry: not this forward's
        ns._handoff_put(imgs[0], (1, 2, 2, 3), (torch.zeros(5000), None, None, 8))   # one entry over the budget: it stays
        assert list(ns._HANDOFF) == [imgs[0].data_ptr()]
        k = imgs[0].data_ptr()
        del imgs
        import gc
        gc.collect()
        assert k not in ns._HANDOFF                                    # the storage's finaliser
    finally:
        ns.HANDOFF_BYTES_MAX = old_max
        ns._HANDOFF.clear()
        ns._HANDOFF.update(old)


@pytest.mark.gpu
def test_shim_order_check():
    """raymarch_forward's sortedobjid must be the fixed identity order (mvpraymarch.py:45).  Identity passes on every call.  A
    permuted order fails INSIDE the offending call when gradients are off (a render may be the only call there is) and during
    the first STRICT_FIRST_CALLS calls; in a training step past those the verdict is read without blocking -- by the same
    step's backward or the next call -- and a tensor is never marked good before its verdict says so: handing the same bad
    tensor in again is refused again.  A wrong shape raises at once."""
    import extensions.mvpraymarch.mvpraymarchlib as lib
    from ava256_amd import native_shim as ns
    from ava256_amd.scene import make_scene
    N, H, W, K = 1, 32, 32, 64
    s = make_scene(N, H, W, K, device="cuda", seed=5)
    import ava256_amd as ops
    rp, rd, tm = ops.compute_raydirs(s["campos"], s["camrot"], s["focal"], s["princpt"], s["pixelcoords"], s["volradius"])
    dev = rp.device
    nodechildren = torch.zeros((N, 2 * K - 1, 2), dtype=torch.int32, device=dev)
    nodeaabb = torch.empty((N, 2 * K - 1, 2, 3), device=dev)
    lib.compute_aabb(s["primpos"], s["primrot"], s["primscale"], None, nodechildren, None, nodeaabb, 0)

    def fwd(order, grad=False):
        rgba = torch.empty((N, H, W, 4), device=dev)
        raysat = torch.empty((N, H, W, 3), device=dev) if grad else None
        lib.raymarch_forward(rp, rd, s["stepsize"], tm, order, nodechildren, nodeaabb, s["primpos"], s["primrot"],
                             s["primscale"], s["template"], None, rgba, raysat, None, 0, False, 512, True, True)
        return rgba

    ns.flush_order_checks()
    ns._VERIFIED.clear()
    ident = lambda: (torch.arange(N * K, dtype=torch.int32, device=dev) % K).view(N, K)
    perm = ident().flip(1).contiguous()
    # (1) without gradients the offending call itself fails, every time, also past the strict first calls -- as long as fewer
    #     than STRICT_FIRST_CALLS identity verdicts have been read for tensors of this kind
    ns._ORDER_CALLS[0] = ns.STRICT_FIRST_CALLS + 100
    for _ in range(2):
        with pytest.raises(NotImplementedError, match="this raymarch call"):
            fwd(perm)
    assert getattr(perm, "_mvp_identity", None) is None          # never marked good
    good = ident()
    fwd(good)
    assert good._mvp_identity == good._version                   # marked once its verdict was read
    # (2) a training step past the strict first calls does not block: the verdict is read later ...
    for _ in range(3):
        fwd(ident(), grad=True)
    ns.flush_order_checks()                                      # nothing to report
    bad2 = ident().flip(1).contiguous()
    fwd(bad2, grad=True)                                         # enqueued, not yet judged
    torch.cuda.synchronize()
    with pytest.raises(NotImplementedError, match="EARLIER"):
        fwd(ident(), grad=True)                                  # ... the next call reports it
    with pytest.raises(NotImplementedError):
        fwd(bad2, grad=True); torch.cuda.synchronize(); ns.flush_order_checks()   # the same bad tensor again: refused again
    ns.flush_order_checks()                                      # (and a report is not repeated)
    # (3) the first calls of a process are strict in grad mode too
    ns._ORDER_CALLS[0] = 0
    with pytest.raises(NotImplementedError, match="this raymarch call"):
        fwd(ident().flip(1).contiguous(), grad=True)
    ns._ORDER_CALLS[0] = ns.STRICT_FIRST_CALLS + 100
    with pytest.raises(NotImplementedError):
        fwd(ident()[:, : K // 2].contiguous())
    # (4) a render LOOP (the reference's glue hands a new sortedobjid to every forward, so no per-tensor cache can hit): once
    #     STRICT_FIRST_CALLS identity verdicts have been read for this kind of tensor, a call without gradients no longer waits
    #     inside the call; a wrong order is then reported by the next call or by flush_order_checks(), and says whose it was
    ns.flush_order_checks()
    ns._VERIFIED.clear()
    for _ in range(ns.STRICT_FIRST_CALLS):
        fwd(ident())                                             # each waits for its verdict inside the call
    assert sum(ns._VERIFIED.values()) == ns.STRICT_FIRST_CALLS and not ns._ORDER_PENDING
    fwd(ident())
    assert len(ns._ORDER_PENDING) == 1                           # deferred: nobody waited
    ns.flush_order_checks()
    fwd(ident().flip(1).contiguous())                            # a bad order now passes the call itself ...
    torch.cuda.synchronize()
    with pytest.raises(NotImplementedError, match="EARLIER"):
        fwd(ident())                                             # ... and the next call reports it
    ns.flush_order_checks()
    ns._VERIFIED.clear()
