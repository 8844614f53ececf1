"""Native-module shims: the reference's two pybind modules, re-created over the C ABI with their POSITIONAL signatures.

The drop-in boundary of this build is the operator layer (`mvpraymarch`, `compute_raydirs`, `Raymarcher`).
This file adds the layer below it: functions with exactly the parameter lists of `mvpraymarchlib`
(extensions/mvpraymarch/mvpraymarch.cpp:146-396 -> `compute_morton, build_tree, compute_aabb, raymarch_forward,
raymarch_backward`, registered at :398-405) and `utilslib` (extensions/utils/utils.cpp:46-82 ->
`compute_raydirs_forward, compute_raydirs_backward`, :134-137), so that the reference's UNMODIFIED
`extensions/mvpraymarch/mvpraymarch.py` and `extensions/utils/utils.py` can keep their `from . import mvpraymarchlib` /
`from . import utilslib`: copy `extensions/mvpraymarch/mvpraymarchlib.py` and `extensions/utils/utilslib.py` of this
repository (two-line re-exports of this module) next to them instead of building the CUDA extensions.

What the shims do with arguments the gfx950 kernels have no use for:
  sortedobjid, nodechildren, nodeparent   ignored: the fixed-order tree is implicit (only usebvh="fixedorder" is valid;
                                          compute_morton / build_tree raise -- the reference's traversal never reads the
                                          LBVH topology they would build, utils.h:742,788)
  sortprims, maxhitboxes, synchitboxes, accum, termthresh, griddim, blocksizex/y, rayterm
                                          ignored like the reference's kernels ignore them (hard-wired template
                                          arguments, mvpraymarch_kernel.cu:33,101-102,188-189)
  chlast = False                          rejected (the reference's sampler is instantiated channels-last only)
The forward keeps the hand-off buffers of the primitive-centric backward (rayaux, packet lists) attached to the STORAGE
of the `rayrgba` tensor the reference's autograd Function saves and passes back (the saved tensor is unpacked as a new
tensor object over the same storage), so the backward runs the fast path; the entry dies with that storage.
"""
import importlib
import weakref

import torch

from . import _hooks, _lib
from ._tensors import aligned, ptr, require_device_f32, stream_ptr

_op = importlib.import_module(__package__ + ".mvpraymarch")  # (the package re-exports a function of that name)

# rayrgba.data_ptr() -> ((N, H, W, K), rayaux, pl_count, pl_list, pl_cap), in order of insertion.  An entry lives at most as
# long as the rayrgba STORAGE (a finaliser on the storage object removes it: a forward whose backward never runs leaks
# nothing, and the caching allocator cannot hand the address to another tensor while the entry exists) -- and the dict is
# bounded in BYTES: a caller that keeps grad-mode images alive (logging, evaluation without no_grad) would otherwise keep
# every forward's hand-off buffers with them.  Over the budget the oldest entries go; a backward that finds none takes the
# ray-centric kernel, which needs no hand-off (correct, slow).
# Threads: the forward shim (host thread) inserts and evicts, the backward shim (autograd thread) only reads with .get(); each
# is a single dict operation under the GIL, and an entry evicted between the two costs a backward the fast path, not its
# correctness (it then takes the ray-centric kernel).
_HANDOFF = {}
HANDOFF_BYTES_MAX = 4 << 30


def _entry_bytes(ent):
    return sum(t.numel() * t.element_size() for t in ent[1:4] if t is not None)


def _handoff_put(rayrgba, shape, buffers):
    key = rayrgba.data_ptr()
    _HANDOFF.pop(key, None)
    _HANDOFF[key] = (shape,) + tuple(buffers)
    weakref.finalize(rayrgba.untyped_storage(), _HANDOFF.pop, key, None)
    total = sum(_entry_bytes(e) for e in _HANDOFF.values())
    for k in list(_HANDOFF):          # oldest first; the entry just added stays
        if total <= HANDOFF_BYTES_MAX or k == key:
            break
        total -= _entry_bytes(_HANDOFF.pop(k))


def _handoff_take(rayrgba, shape):
    """The buffers of the grad-mode forward that wrote `rayrgba`, or Nones (-> ray-centric backward) when there was
    none (or it was evicted) or its geometry is not this call's.  Left in place: a second backward over the same forward
    (retain_graph) finds them again; the storage's finaliser removes them."""
    ent = _HANDOFF.get(rayrgba.data_ptr())
    if ent is None or ent[0] != shape:
        return None, None, None, 0
    return ent[1:]


STRICT_ORDER_CHECK = False   # True: every call validates `sortedobjid` with a host synchronisation inside the call
STRICT_FIRST_CALLS = 8       # ... as the first calls of a process do anyway, and every call that renders without gradients
_ORDER_PENDING = []          # [(event, pinned word, weakref(tensor), version)]: device-side checks not yet looked at
_ORDER_CALLS = [0]           # content checks enqueued so far
_ARANGE = {}                 # (device index, K, dtype) -> arange(K)
_VERIFIED = {}               # (device index, K, dtype) -> verdicts "identity" read so far for that kind of tensor
_BAD_ORDER = ("sortedobjid of %s raymarch call is not the fixed identity order (its results are in the wrong composition "
              "order): only usebvh='fixedorder' without randomorder is supported")


def _poll_order_checks(wait=False, current=None):
    """Look at the device-side verdicts that have arrived (all of them with wait=True).  A tensor is marked as checked only
    HERE, once its verdict has been read as "identity" -- never before the verdict is known, so a caller that catches the
    error and hands the same bad tensor in again is refused again.  Raises for a bad order; the other pending verdicts stay
    queued.  Called from the forward and backward shims (host thread and autograd thread): the list is swapped, not edited in
    place, and a verdict looked at twice is harmless."""
    pending, bad, bad_current, keep = list(_ORDER_PENDING), False, False, []
    for item in pending:
        ev, word, ref, ver, key = item
        if wait:
            ev.synchronize()
        if ev.query():
            if int(word[0]) != 0:
                bad = True
                bad_current = bad_current or item is current
            else:
                _VERIFIED[key] = _VERIFIED.get(key, 0) + 1
                t = ref()
                if t is not None and t._version == ver:
                    t._mvp_identity = ver
        else:
            keep.append(item)
    _ORDER_PENDING[:] = keep + [i for i in _ORDER_PENDING if i not in pending]
    if bad:   # (`current` = the check the calling shim has just queued: the message says whose tensor it was)
        raise NotImplementedError(_BAD_ORDER % ("this" if bad_current else "an EARLIER"))


def flush_order_checks():
    """Public: wait for every outstanding `sortedobjid` verdict and raise if one of them was a wrong order.  A caller that
    renders once and reads the image back (evaluation) may call this before trusting the result of a deferred check; the
    shim itself checks INSIDE the call whenever gradients are off, and for the first STRICT_FIRST_CALLS calls."""
    _poll_order_checks(wait=True)


def _identity_order(sortedobjid, K, strict=False):
    """usebvh='fixedorder' hands arange(K) per image (mvpraymarch.py:45); any other order (randomorder=True, the LBVH
    path) would silently render in the wrong composition order here.  Shape errors raise at once.  The CONTENT is compared
    on the device; WHEN the one-word verdict is read depends on who is calling:
      * the first STRICT_FIRST_CALLS calls of the process, everything under STRICT_ORDER_CHECK, and a call without gradients
        (a render / evaluation: possibly the only call there is) UNTIL STRICT_FIRST_CALLS verdicts "identity" have been read
        for tensors of its kind (device, K, dtype) wait for it inside the call -- the offending call itself fails, before its
        image can be used.  After that a render loop is treated like a training loop: the reference's glue makes a new
        sortedobjid per forward (mvpraymarch.py:45), so a per-tensor cache can never hit and waiting would cost every render a
        host synchronisation; `flush_order_checks()` is there for a caller that wants the verdict before using an image;
      * a training step (the reference's glue builds a NEW sortedobjid on every forward, so the check runs on every call
        and must not block the host) leaves the verdict behind an event; the backward of the same step and the next calls
        look at it.  An order policy is a constant of a run, so a wrong one has failed within the strict first calls.
    Nothing is enqueued while a stream is being captured."""
    if sortedobjid is None:
        return
    _poll_order_checks()
    if sortedobjid.dim() != 2 or sortedobjid.size(1) != K:
        raise NotImplementedError("sortedobjid must be [N, K] in the fixed identity order (usebvh='fixedorder')")
    if getattr(sortedobjid, "_mvp_identity", None) == sortedobjid._version or torch.cuda.is_current_stream_capturing():
        return
    dev = sortedobjid.device
    key = (dev.index, K, sortedobjid.dtype)
    ar = _ARANGE.get(key)
    if ar is None:
        ar = _ARANGE[key] = torch.arange(K, device=dev, dtype=sortedobjid.dtype)
    word = torch.empty(1, dtype=torch.int32, pin_memory=True)
    word.copy_((sortedobjid != ar[None]).any().to(torch.int32).reshape(1), non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))
    item = (ev, word, weakref.ref(sortedobjid), sortedobjid._version, key)
    _ORDER_PENDING.append(item)
    _ORDER_CALLS[0] += 1
    if ((strict and _VERIFIED.get(key, 0) < STRICT_FIRST_CALLS) or STRICT_ORDER_CHECK or _ORDER_CALLS[0] <= STRICT_FIRST_CALLS
            or len(_ORDER_PENDING) > 64):
        _poll_order_checks(wait=True, current=item)


def compute_morton(*args):
    raise NotImplementedError("compute_morton: only usebvh='fixedorder' is supported (the reference's traversal "
                              "ignores the LBVH topology, utils.h:742,788)")


def build_tree(*args):
    raise NotImplementedError("build_tree: only usebvh='fixedorder' is supported")


def compute_aabb(primpos, primrot, primscale, sortedobjid, nodechildren, nodeparent, nodeaabb, algo):
    """mvpraymarch.cpp:198-230 -> mvp_aabb_build."""
    primpos, primrot, primscale = (require_device_f32(n, t) for n, t in
                                   (("primpos", primpos), ("primrot", primrot), ("primscale", primscale)))
    N, K = primpos.size(0), primpos.size(1)
    assert nodeaabb.is_contiguous() and tuple(nodeaabb.shape) == (N, 2 * K - 1, 2, 3) and nodeaabb.dtype == torch.float32
    dev = primpos.device
    with torch.cuda.device(dev):
        _lib.check(_lib.get_lib().mvp_aabb_build(N, K, ptr(primpos), ptr(primrot), ptr(primscale), ptr(nodeaabb),
                                                stream_ptr(dev)), "mvp_aabb_build")


def raymarch_forward(raypos, raydir, stepsize, tminmax, sortedobjid, nodechildren, nodeaabb, primpos, primrot,
                     primscale, template, warp, rayrgba, raysat, rayterm, algo=0, sortprims=True, maxhitboxes=512,
                     synchitboxes=False, chlast=False, fadescale=8.0, fadeexp=8.0, accum=0, termthresh=0.0, griddim=3,
                     blocksizex=8, blocksizey=16):
    """mvpraymarch.cpp:180-280 (same defaults) -> mvp_march_forward.  `raysat` given = grad mode (mvpraymarch.py:147-152)."""
    if not chlast:
        raise NotImplementedError("channels-first templates: the reference instantiates only the channels-last sampler")
    if algo not in (0, 1):
        raise NotImplementedError("algo must be 0 or 1")
    for n, t in (("raypos", raypos), ("raydir", raydir), ("tminmax", tminmax), ("nodeaabb", nodeaabb),
                 ("primpos", primpos), ("primrot", primrot), ("primscale", primscale), ("template", template)):
        require_device_f32(n, t)
    N, H, W = raypos.size(0), raypos.size(1), raypos.size(2)
    K = primpos.size(1)
    TD, TH, TW = template.size(2), template.size(3), template.size(4)
    dev = raypos.device
    if algo == 0:
        warp = None
    WD, WH, WW = (warp.size(2), warp.size(3), warp.size(4)) if warp is not None else (0, 0, 0)
    _identity_order(sortedobjid, K, strict=raysat is None)   # (no raysat = no gradients: mvpraymarch.py:147-152)
    raypos, raydir, tminmax, template = aligned(raypos), aligned(raydir), aligned(tminmax), aligned(template)
    if warp is not None:
        warp = aligned(require_device_f32("warp", warp))
    require_device_f32("rayrgba", rayrgba)
    if rayrgba.data_ptr() & 15:
        raise RuntimeError("rayrgba must be 16-byte aligned")
    rayaux = pl_count = pl_list = None
    pl_cap = 0
    if raysat is not None:
        require_device_f32("raysat", raysat)
        rayaux, pl_count, pl_list, pl_cap = _op.alloc_handoff(N, H, W, K, dev)
        _handoff_put(rayrgba, (N, H, W, K), (rayaux, pl_count, pl_list, pl_cap))
    with torch.cuda.device(dev):
        _lib.check(_lib.get_lib().mvp_march_forward(
            N, H, W, K, ptr(raypos), ptr(raydir), float(stepsize), ptr(tminmax), ptr(nodeaabb), ptr(primpos),
            ptr(primrot), ptr(primscale), TD, TH, TW, ptr(template), WD, WH, WW, ptr(warp), ptr(rayrgba), ptr(raysat),
            ptr(rayaux), ptr(pl_count), ptr(pl_list), pl_cap, float(fadescale), float(fadeexp), ptr(_hooks.diag),
            stream_ptr(dev)), "mvp_march_forward")
        if pl_count is not None:
            _op.note_list_demand(pl_count, N, H, W, K)


def raymarch_backward(raypos, raydir, stepsize, tminmax, sortedobjid, nodechildren, nodeaabb, primpos, grad_primpos,
                      primrot, grad_primrot, primscale, grad_primscale, template, grad_template, warp, grad_warp,
                      rayrgba, grad_rayrgba, raysat, rayterm, algo=0, sortprims=True, maxhitboxes=512, synchitboxes=False,
                      chlast=False, fadescale=8.0, fadeexp=8.0, accum=0, termthresh=0.0, griddim=3, blocksizex=8,
                      blocksizey=16):
    """mvpraymarch.cpp:282-396 (same defaults) -> mvp_march_backward (the gradient buffers are overwritten; the zeros the reference
    pre-fills them with, mvpraymarch.py:240-246, are not needed)."""
    if not chlast:
        raise NotImplementedError("channels-first templates: the reference instantiates only the channels-last sampler")
    if algo not in (0, 1):
        raise NotImplementedError("algo must be 0 or 1")
    # the same checks and 16-byte treatment as the forward (an offset view must not pass one and fail the other)
    for n, t in (("raypos", raypos), ("raydir", raydir), ("tminmax", tminmax), ("nodeaabb", nodeaabb),
                 ("primpos", primpos), ("primrot", primrot), ("primscale", primscale), ("template", template),
                 ("raysat", raysat), ("grad_primpos", grad_primpos), ("grad_primrot", grad_primrot),
                 ("grad_primscale", grad_primscale), ("grad_template", grad_template)):
        require_device_f32(n, t)
    N, H, W = raypos.size(0), raypos.size(1), raypos.size(2)
    K = primpos.size(1)
    TD, TH, TW = template.size(2), template.size(3), template.size(4)
    dev = raypos.device
    _identity_order(sortedobjid, K)
    if algo == 0:
        warp = grad_warp = None
    WD, WH, WW = (warp.size(2), warp.size(3), warp.size(4)) if warp is not None else (0, 0, 0)
    raypos, raydir, tminmax, template = aligned(raypos), aligned(raydir), aligned(tminmax), aligned(template)
    if warp is not None:
        warp = aligned(require_device_f32("warp", warp))
    if grad_template.data_ptr() & 15:
        raise RuntimeError("grad_template must be 16-byte aligned")  # an output: cannot be replaced by a copy
    rayaux, pl_count, pl_list, pl_cap = _handoff_take(rayrgba, (N, H, W, K))
    grad_rayrgba = aligned(require_device_f32("grad_rayrgba", grad_rayrgba))
    with torch.cuda.device(dev):
        _lib.check(_lib.get_lib().mvp_march_backward(
            N, H, W, K, ptr(raypos), ptr(raydir), float(stepsize), ptr(tminmax), ptr(nodeaabb), ptr(primpos),
            ptr(primrot), ptr(primscale), TD, TH, TW, ptr(template), WD, WH, WW, ptr(warp), ptr(raysat), ptr(rayaux),
            ptr(pl_count), ptr(pl_list), pl_cap, ptr(grad_rayrgba), ptr(grad_primpos), ptr(grad_primrot),
            ptr(grad_primscale), ptr(grad_template), ptr(grad_warp), float(fadescale), float(fadeexp), ptr(_hooks.diag),
            stream_ptr(dev)), "mvp_march_backward")


def compute_raydirs_forward(viewpos, viewrot, focal, princpt, pixelcoords, W, H, volradius, raypos, raydir, tminmax):
    """utils.cpp:46-64 -> mvp_raydirs_forward (pixelcoords may be None: integer pixel grid)."""
    for n, t in (("viewpos", viewpos), ("viewrot", viewrot), ("focal", focal), ("princpt", princpt)):
        require_device_f32(n, t)
    for n, t in (("raypos", raypos), ("raydir", raydir), ("tminmax", tminmax)):
        require_device_f32(n, t)
    N = viewpos.size(0)
    dev = viewpos.device
    pc = aligned(require_device_f32("pixelcoords", pixelcoords)) if pixelcoords is not None else None
    with torch.cuda.device(dev):
        _lib.check(_lib.get_lib().mvp_raydirs_forward(N, int(H), int(W), ptr(viewpos), ptr(viewrot), ptr(focal),
                                                     ptr(princpt), ptr(pc), float(volradius), ptr(raypos), ptr(raydir),
                                                     ptr(tminmax), stream_ptr(dev)), "mvp_raydirs_forward")


def compute_raydirs_backward(*args):
    """utils.cpp:66-82: the reference's backward kernel writes nothing (utils_kernel.cu:54-95); neither does this."""
    return None
