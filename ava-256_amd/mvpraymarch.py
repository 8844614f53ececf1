"""mvpraymarch -- the MVP raymarch operator with the reference's Python surface
(extensions/mvpraymarch/mvpraymarch.py:21-84 build_accel, :87-292 MVPRaymarch, :295-390 mvpraymarch),
running the gfx950 kernels of csrc/{aabb,march}.hip through the C ABI (include/mvp_abi.h).

Differences that are deliberate:
  * the "fixedorder" tree is an implicit heap, so no sortedobjid / nodechildren / nodeparent tensors are
    built (the reference builds them with ~10 small torch kernels per call, mvpraymarch.py:44-75);
  * kernels run on the CURRENT stream, not on legacy stream 0, and nothing is allocated inside the library;
  * options the reference's kernels ignore (sortprims, maxhitboxes, synchitboxes, accum, termthresh, griddim,
    blocksize, bwdblocksize: hard-wired template arguments, mvpraymarch_kernel.cu:33,101-102,188-189) are
    accepted and ignored here too; options that would select code this build does not have raise.
"""
import torch
from torch.autograd import Function

from . import _hooks, _lib
from ._tensors import aligned, ptr, require_device_f32, stream_ptr

class _ListDemand:
    """What the forward's per-primitive counters said the lists of one problem shape need (they keep counting past the
    capacity): a decaying estimate of the wanted capacity, the capacity in use, and one measurement in flight (device
    histogram -> pinned words -> event)."""
    __slots__ = ("demand", "cap", "hist", "words", "event")

    def __init__(self):
        self.demand, self.cap, self.hist, self.words, self.event = 0.0, 0, None, None, None

    def poll(self):
        if self.event is not None and self.event.query():
            self.note(wanted_from_histogram(self.words.tolist()))
            self.event = None

    def note(self, wanted):
        """One measurement.  The estimate follows a rise at once and forgets it at 10 % per measurement: a close-up
        iteration (or an exploding scale early in training) costs its own iterations, not the rest of the run."""
        self.demand = max(float(wanted), 0.9 * self.demand)


def wanted_from_histogram(words):
    """Capacity one measurement asks for: the largest count any primitive showed -- unless that is an outlier.  `words` =
    mvp_list_demand's 256 bins of width 8 (counts clamped to 2047) + the maximum.  When the maximum is more than twice the
    99.9th percentile (one image-filling primitive among thousands), the lists are sized for 2 x that percentile and
    the few primitives above it stay on the ray-centric kernel, which marches only their packets: sizing N*K lists for one
    primitive would cost gigabytes (N*K*cap*16 bytes; C2: 5.2 MB per unit of capacity)."""
    hist, mx = words[:256], int(words[256])
    n = sum(hist)
    if n == 0:
        return 0
    allowed, acc, q = n // 1000, 0, 0       # primitives allowed above the percentile
    for b in range(255, -1, -1):
        acc += hist[b]
        if acc > allowed:
            q = 8 * b + 7                   # upper edge of the bin that holds the percentile
            break
    return mx if mx <= 2 * max(q, 16) else 2 * max(q, 16)


# Written by forwards only (host thread; the backward on the autograd thread never touches it): single dict operations
# under the GIL, and a lost update costs one call a stale capacity, never a wrong result (lists over capacity are marched by
# the ray-centric kernel).
_LIST_DEMAND = {}  # (device index, H, W, K) -> _ListDemand
LIST_ENTRY_WORDS = 4        # uint32 words per list entry (include/mvp_abi.h: key, step range, 64-bit ray mask)
_LIST_BYTES_MIN = 128 << 20  # the lists of a call may take this much ...
_LIST_BYTES_PER_PRIM = 4096  # ... or this much per primitive (half an 8^3 slab = 256 entries), whichever is more


def _budget_clamp(cap, N, K):
    """The memory budget of the lists: max(128 MiB, 4 KiB per primitive), never below 32 entries."""
    if N is not None and N * K > 0:
        budget = max(_LIST_BYTES_MIN, _LIST_BYTES_PER_PRIM * N * K)
        cap = max(32, min(cap, budget // (4 * LIST_ENTRY_WORDS * N * K) // 8 * 8))
    return cap


def capacity_for_demand(demand, N, K):
    """The capacity the feedback asks for at a measured demand (no hysteresis): 1.25 x, a multiple of 8 in [32, 2048], inside
    the memory budget."""
    return _budget_clamp((int(min(max(32.0, 1.25 * float(demand)), 2048)) + 7) // 8 * 8, N, K)


def capacity_wanted_now(pl_count, N, H, W, K):
    """What capacity would the feedback choose for the counters in `pl_count` (a forward's hand-off words)?  SYNCHRONOUS
    (histogram launch + 1 KB read-back): for the rare look a captured training graph takes at its frozen capacity
    (trainloop.Trainer.step).  The measurement is also noted for the shape, so that eager calls that follow size their lists
    from it at once."""
    dev = pl_count.device
    hist = torch.empty(257, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.get_lib().mvp_list_demand(ptr(pl_count), N * K, ptr(hist), stream_ptr(dev)), "mvp_list_demand")
    wanted = wanted_from_histogram(hist.cpu().tolist())
    _LIST_DEMAND.setdefault((dev.index, H, W, K), _ListDemand()).note(wanted)
    return capacity_for_demand(wanted, N, K)


def primlist_capacity(H, W, K, device=None, N=None):
    """Per-primitive capacity of the packet lists handed from forward to backward.  First call of a shape: a heuristic
    -- on head-like scenes a packet (8x8 pixels) lists ~19 primitives and ~46 % of the packets hit anything (measured, C2),
    so a primitive is listed by ~9-10 * packets / K packets; four times that, at least 32.  Afterwards: 1.25 x the measured
    demand (`note_list_demand`: read one call late and without a host synchronisation; outliers clipped, decaying --
    `_ListDemand.note`, `wanted_from_histogram`), kept while the new value is within [0.6, 1] of the one in use so that
    the allocation size does not flutter.  Multiple of 8 (the library reads lists 32 bytes at a time), at most 2048, and --
    when N is given -- at most what a memory budget of max(128 MiB, 4 KiB per primitive) allows, never below 32: primitives
    over the capacity are handled by the ray-centric kernel, correctly and slowly."""
    packets = ((H + 7) // 8) * ((W + 7) // 8)
    cap = (int(min(max(32.0, 4 * 10.0 * packets / max(K, 1)), 2048)) + 7) // 8 * 8
    st = _LIST_DEMAND.get((getattr(device, "index", None), H, W, K)) if device is not None else None
    if st is not None:
        if st.event is not None and not torch.cuda.is_current_stream_capturing():   # (an event query cannot be captured)
            st.poll()
        if st.demand > 0:
            cap = (int(min(max(32.0, 1.25 * st.demand), 2048)) + 7) // 8 * 8
            if st.cap and 0.6 * st.cap <= cap <= st.cap:
                cap = st.cap
            st.cap = cap
    return _budget_clamp(cap, N, K)


def note_list_demand(pl_count, N, H, W, K):
    """Queue a read-back of the demand statistics of the counters the forward has just written (stream-ordered behind it,
    before any backward marks them): one histogram launch (mvp_list_demand), a 1 KB copy into pinned memory and an event.
    At most one in flight per shape; nothing happens while a stream is being captured."""
    dev = pl_count.device
    if N * K == 0 or torch.cuda.is_current_stream_capturing():
        return
    st = _LIST_DEMAND.setdefault((dev.index, H, W, K), _ListDemand())
    st.poll()
    if st.event is not None:
        return
    if st.words is None:
        st.words = torch.zeros(257, dtype=torch.int32, pin_memory=True)
        st.hist = torch.empty(257, dtype=torch.int32, device=dev)
    _lib.check(_lib.get_lib().mvp_list_demand(ptr(pl_count), N * K, ptr(st.hist), stream_ptr(dev)), "mvp_list_demand")
    st.words.copy_(st.hist, non_blocking=True)
    st.event = torch.cuda.Event()
    st.event.record(torch.cuda.current_stream(dev))


def alloc_handoff(N, H, W, K, dev):
    """Hand-off buffers of the primitive-centric backward (include/mvp_abi.h): per-ray saturation record, per-primitive
    counters + flags + per-packet words (zeroed by the library), and per primitive the list of ray packets that touch it.
    Returns (rayaux, pl_count, pl_list, pl_cap)."""
    pl_cap = primlist_capacity(H, W, K, dev, N)
    rayaux = torch.empty((N, H, W, 4), device=dev, dtype=torch.int32)
    pl_count = torch.empty((N * K + 3 + N * ((H + 7) // 8) * ((W + 7) // 8),), device=dev, dtype=torch.int32)
    pl_list = torch.empty((N * K, pl_cap, LIST_ENTRY_WORDS), device=dev, dtype=torch.int32)
    return rayaux, pl_count, pl_list, pl_cap


def build_accel(primtransfin, algo, fixedorder=False):
    """AABBs of the fixed-order heap BVH.  Returns (sortedobjid, nodechildren, nodeaabb) like the reference
    (mvpraymarch.py:84); the first two are None because the topology is implicit (leaf K-1+k = primitive k,
    children of i are 2i+1 / 2i+2 -- exactly the tensors mvpraymarch.py:45,57-70 would spell out)."""
    if not fixedorder:
        raise NotImplementedError(
            "only usebvh='fixedorder' is implemented: the reference's traversal ignores the LBVH topology "
            "(utils.h:742,788), so its usebvh=True path walks the wrong tree")
    primpos, primrot, primscale = primtransfin
    N, K = primpos.size(0), primpos.size(1)
    dev = primpos.device
    nodeaabb = torch.empty((N, K + K - 1, 2, 3), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev), _hooks.timed("aabb_build", dev):
        _lib.check(_lib.get_lib().mvp_aabb_build(N, K, ptr(primpos), ptr(primrot), ptr(primscale), ptr(nodeaabb),
                                                stream_ptr(dev)), "mvp_aabb_build")
    return None, None, nodeaabb


def _forward_impl(ctx, rays, cams, stepsize, primpos, primrot, primscale, template, warp, gradmode, options):
    """Shared body of MVPRaymarch.forward (rays = (raypos, raydir, tminmax)) and MVPRaymarchFromCameras.forward
    (cams = (campos, camrot, focal, princpt, pixelcoords-or-(W,H), volradius): the rays are made inside the march)."""
    algo = options["algo"]
    usebvh = options["usebvh"]
    if algo not in (0, 1):
        raise NotImplementedError("algo must be 0 (slab sampler) or 1 (warp-field sampler)")
    if algo == 1 and warp is None:
        raise RuntimeError("algo=1 needs a warp field")
    if algo == 0:
        warp = None  # PrimSamplerTW<false> never reads it (mvpraymarch_kernel.cu:89-95)
    if usebvh != "fixedorder":
        raise NotImplementedError("only usebvh='fixedorder' is implemented")
    if options.get("randomorder", False):
        raise NotImplementedError("randomorder is not implemented (the reference indexes dim 0 there, "
                                  "mvpraymarch.py:139-140)")
    if not options.get("chlast", True):
        raise NotImplementedError("MVPRaymarch.forward takes channels-last templates; use mvpraymarch(chlast=False)")
    fadescale, fadeexp = float(options["fadescale"]), float(options["fadeexp"])

    primpos = require_device_f32("primpos", primpos)
    primrot = require_device_f32("primrot", primrot)
    primscale = require_device_f32("primscale", primscale)
    template = require_device_f32("template", template)
    if cams is None:
        raypos, raydir, tminmax = rays
        raypos = require_device_f32("raypos", raypos)
        raydir = require_device_f32("raydir", raydir)
        tminmax = require_device_f32("tminmax", tminmax)
        assert raypos.dim() == 4 and raypos.size(3) == 3
        assert raydir.shape == raypos.shape
        assert tminmax.shape == raypos.shape[:3] + (2,)
        N, H, W = raypos.size(0), raypos.size(1), raypos.size(2)
        raypos, raydir, tminmax = aligned(raypos), aligned(raydir), aligned(tminmax)
    else:
        if warp is not None:
            raise NotImplementedError("the fused camera entry point has no warp-field variant (algo 0 only)")
        campos, camrot, focal, princpt, pixelcoords, volradius = cams
        campos, camrot = require_device_f32("campos", campos), require_device_f32("camrot", camrot)
        focal, princpt = require_device_f32("focal", focal), require_device_f32("princpt", princpt)
        N = campos.size(0)
        if isinstance(pixelcoords, tuple):
            W, H = pixelcoords
            pc = None
        else:
            pc = aligned(require_device_f32("pixelcoords", pixelcoords))
            H, W = pc.size(1), pc.size(2)
            assert pc.size(0) == N and pc.size(3) == 2
        assert campos.shape == (N, 3) and camrot.shape == (N, 3, 3) and focal.shape == (N, 2) and princpt.shape == (N, 2)
        raypos = raydir = tminmax = None
    K = primpos.size(1)
    assert primpos.shape == (N, K, 3) and primrot.shape == (N, K, 3, 3) and primscale.shape == (N, K, 3)
    assert template.dim() == 6 and template.size(-1) == 4 and template.shape[:2] == (N, K)
    TD, TH, TW = template.size(2), template.size(3), template.size(4)
    dev = primpos.device
    WD = WH = WW = 0
    if warp is not None:
        warp = aligned(require_device_f32("warp", warp))
        assert warp.dim() == 6 and warp.size(-1) == 3 and warp.shape[:2] == (N, K)   # mvpraymarch.py:124
        WD, WH, WW = warp.size(2), warp.size(3), warp.size(4)

    template = aligned(template)
    _, _, nodeaabb = build_accel((primpos, primrot, primscale), algo, fixedorder=True)

    rayrgba = torch.empty((N, H, W, 4), device=dev, dtype=torch.float32)
    raysat = rayaux = pl_count = pl_list = None
    pl_cap = 0
    if gradmode:
        raysat = torch.empty((N, H, W, 3), device=dev, dtype=torch.float32)
        rayaux, pl_count, pl_list, pl_cap = alloc_handoff(N, H, W, K, dev)
    with torch.cuda.device(dev), _hooks.timed("march_forward", dev):
        if cams is None:
            _lib.check(_lib.get_lib().mvp_march_forward(
                N, H, W, K, ptr(raypos), ptr(raydir), float(stepsize), ptr(tminmax), ptr(nodeaabb), ptr(primpos),
                ptr(primrot), ptr(primscale), TD, TH, TW, ptr(template), WD, WH, WW, ptr(warp), ptr(rayrgba),
                ptr(raysat), ptr(rayaux), ptr(pl_count), ptr(pl_list), pl_cap, fadescale, fadeexp, ptr(_hooks.diag),
                stream_ptr(dev)), "mvp_march_forward")
        else:
            if gradmode:  # the backward's ray tensors: written by the forward march itself (no raydirs launch)
                raypos = torch.empty((N, H, W, 3), device=dev, dtype=torch.float32)
                raydir = torch.empty((N, H, W, 3), device=dev, dtype=torch.float32)
                tminmax = torch.empty((N, H, W, 2), device=dev, dtype=torch.float32)
            _lib.check(_lib.get_lib().mvp_march_forward_cams(
                N, H, W, K, ptr(campos), ptr(camrot), ptr(focal), ptr(princpt), ptr(pc), float(volradius),
                float(stepsize), ptr(nodeaabb), ptr(primpos), ptr(primrot), ptr(primscale), TD, TH, TW, ptr(template),
                ptr(rayrgba), ptr(raysat), ptr(rayaux), ptr(pl_count), ptr(pl_list), pl_cap, ptr(raypos), ptr(raydir),
                ptr(tminmax), fadescale, fadeexp, ptr(_hooks.diag), stream_ptr(dev)), "mvp_march_forward_cams")

    if pl_count is not None:
        with torch.cuda.device(dev):
            note_list_demand(pl_count, N, H, W, K)
    if _hooks.keep_raysat:
        _hooks.last_raysat = raysat
        _hooks.last_pl_count = pl_count
        _hooks.last_flags_index = N * K     # pl_count[N*K] = the flags word (include/mvp_abi.h)
        _hooks.last_handoff_shape = (N, H, W, K, pl_cap)
    # (camera form in grad mode: raypos / raydir / tminmax are the tensors the forward march has just written)
    ctx.save_for_backward(raypos, raydir, tminmax, nodeaabb, primpos, primrot, primscale, template, raysat, rayaux,
                          pl_count, pl_list, warp)
    ctx.pl_cap = pl_cap
    ctx.options = options
    ctx.stepsize = float(stepsize)
    return rayrgba


def _backward_impl(ctx, grad_rayrgba):
    (raypos, raydir, tminmax, nodeaabb, primpos, primrot, primscale, template, raysat, rayaux, pl_count,
     pl_list, warp) = ctx.saved_tensors
    if raysat is None:
        raise RuntimeError("backward through mvpraymarch needs grad mode enabled during the forward call")
    fadescale, fadeexp = float(ctx.options["fadescale"]), float(ctx.options["fadeexp"])
    N, H, W = raypos.size(0), raypos.size(1), raypos.size(2)
    K = primpos.size(1)
    TD, TH, TW = template.size(2), template.size(3), template.size(4)
    dev = raypos.device
    grad_rayrgba = aligned(grad_rayrgba.contiguous().float())

    # the library overwrites every element of the four gradients: no zero-fill pass (the reference needs
    # torch.zeros_like x4 here, mvpraymarch.py:240-246)
    grad_primpos = torch.empty_like(primpos)
    grad_primrot = torch.empty_like(primrot)
    grad_primscale = torch.empty_like(primscale)
    grad_template = torch.empty_like(template)
    grad_warp = torch.empty_like(warp) if warp is not None else None
    WD, WH, WW = (warp.size(2), warp.size(3), warp.size(4)) if warp is not None else (0, 0, 0)
    with torch.cuda.device(dev), _hooks.timed("march_backward", dev):
        _lib.check(_lib.get_lib().mvp_march_backward(
            N, H, W, K, ptr(raypos), ptr(raydir), ctx.stepsize, ptr(tminmax), ptr(nodeaabb), ptr(primpos),
            ptr(primrot), ptr(primscale), TD, TH, TW, ptr(template), WD, WH, WW, ptr(warp), ptr(raysat),
            ptr(rayaux), ptr(pl_count), ptr(pl_list), ctx.pl_cap, ptr(grad_rayrgba), ptr(grad_primpos),
            ptr(grad_primrot), ptr(grad_primscale), ptr(grad_template), ptr(grad_warp), fadescale, fadeexp,
            ptr(_hooks.diag), stream_ptr(dev)), "mvp_march_backward")
    return grad_primpos, grad_primrot, grad_primscale, grad_template, grad_warp


class MVPRaymarch(Function):
    """Custom Function for raymarching Mixture of Volumetric Primitives (same argument list as the reference's
    MVPRaymarch.forward, mvpraymarch.py:91-93)."""

    @staticmethod
    def forward(ctx, raypos, raydir, stepsize, tminmax, primpos, primrot, primscale, template, warp, rayterm,
                gradmode, options):
        return _forward_impl(ctx, (raypos, raydir, tminmax), None, stepsize, primpos, primrot, primscale, template,
                             warp, gradmode, options)

    @staticmethod
    def backward(ctx, grad_rayrgba):
        gp, gr, gs, gt, gw = _backward_impl(ctx, grad_rayrgba)
        return (None, None, None, None, gp, gr, gs, gt, gw, None, None, None)


class MVPRaymarchFromCameras(Function):
    """The same operator with the rays made inside the march kernel (mvp_march_forward_cams): the caller's
    compute_raydirs + mvpraymarch pair (models/autoencoder.py:240-252) as one call.  Without gradients no ray tensor
    exists in HBM; in grad mode the forward march writes the rays it made for the backward (what compute_raydirs would
    have written), so a training step has no raydirs launch and its forward reads no ray tensor."""

    @staticmethod
    def forward(ctx, campos, camrot, focal, princpt, pixelcoords, volradius, stepsize, primpos, primrot, primscale,
                template, gradmode, options):
        return _forward_impl(ctx, None, (campos, camrot, focal, princpt, pixelcoords, volradius), stepsize, primpos,
                             primrot, primscale, template, None, gradmode, options)

    @staticmethod
    def backward(ctx, grad_rayrgba):
        gp, gr, gs, gt, _ = _backward_impl(ctx, grad_rayrgba)
        return (None, None, None, None, None, None, None, gp, gr, gs, gt, None, None)


_DEFAULT_OPTIONS = {"algo": 0, "usebvh": "fixedorder", "sortprims": False, "randomorder": False, "maxhitboxes": 512,
                    "synchitboxes": True, "chlast": True, "fadescale": 8.0, "fadeexp": 8.0, "accum": 0,
                    "termthresh": 0.0, "griddim": 3, "blocksize": (8, 16), "bwdblocksize": (8, 16)}


def mvpraymarch_from_cameras(campos, camrot, focal, princpt, pixelcoords, volradius, stepsize, primtransf, template,
                             **options):
    """compute_raydirs(...) + mvpraymarch(...) in one call: same result bit for bit, without the
    [N,H,W,3] + [N,H,W,3] + [N,H,W,2] ray tensors.  Options: the keyword arguments of mvpraymarch (algo 0 only).
    pixelcoords: [N,H,W,2] tensor or a (W, H) tuple like compute_raydirs."""
    unknown = set(options) - set(_DEFAULT_OPTIONS)
    if unknown:
        raise TypeError("unknown mvpraymarch option(s): %s" % sorted(unknown))
    opts = dict(_DEFAULT_OPTIONS)
    opts.update(options)
    if not opts["chlast"]:
        template = template.permute(0, 1, 3, 4, 5, 2).contiguous()
        opts["chlast"] = True
    if isinstance(primtransf, tuple):
        primpos, primrot, primscale = primtransf
    else:
        primpos, primrot, primscale = (primtransf[:, :, 0, :].contiguous(), primtransf[:, :, 1:4, :].contiguous(),
                                       primtransf[:, :, 4, :].contiguous())
    return MVPRaymarchFromCameras.apply(campos, camrot, focal, princpt, pixelcoords, volradius, stepsize, primpos,
                                        primrot, primscale, template, torch.is_grad_enabled(), opts)


def mvpraymarch(
    raypos,
    raydir,
    stepsize,
    tminmax,
    primtransf,
    template,
    warp,
    rayterm=None,
    algo=0,
    usebvh="fixedorder",
    sortprims=False,
    randomorder=False,
    maxhitboxes=512,
    synchitboxes=True,
    chlast=True,
    fadescale=8.0,
    fadeexp=8.0,
    accum=0,
    termthresh=0.0,
    griddim=3,
    blocksize=(8, 16),
    bwdblocksize=(8, 16),
):
    """Main entry point for raymarching MVP; parameter names and defaults are the reference's
    (mvpraymarch.py:295-318) because Raymarcher filters renderoptions by mvpraymarch.__code__.co_varnames.

    raypos, raydir: [N,H,W,3]; tminmax: [N,H,W,2]; primtransf: (primpos [N,K,3], primrot [N,K,3,3],
    primscale [N,K,3]) or packed [N,K,5,3]; template: [N,K,TD,TH,TW,4] (chlast=True) or [N,K,4,TD,TH,TW];
    returns rayrgba [N,H,W,4]."""
    if isinstance(primtransf, tuple):
        primpos, primrot, primscale = primtransf
    else:  # packed layout, mvpraymarch.py:355-360
        primpos, primrot, primscale = (
            primtransf[:, :, 0, :].contiguous(),
            primtransf[:, :, 1:4, :].contiguous(),
            primtransf[:, :, 4, :].contiguous(),
        )
    if not chlast:
        # the reference's kernels only ever read channels-last slabs; do the layout change here, where autograd
        # carries the gradient back to the caller's layout
        template = template.permute(0, 1, 3, 4, 5, 2).contiguous()
        if warp is not None:
            warp = warp.permute(0, 1, 3, 4, 5, 2).contiguous()

    out = MVPRaymarch.apply(
        raypos,
        raydir,
        stepsize,
        tminmax,
        primpos,
        primrot,
        primscale,
        template,
        warp,
        rayterm,
        torch.is_grad_enabled(),
        {
            "algo": algo,
            "usebvh": usebvh,
            "sortprims": sortprims,
            "randomorder": randomorder,
            "maxhitboxes": maxhitboxes,
            "synchitboxes": synchitboxes,
            "chlast": True,
            "fadescale": fadescale,
            "fadeexp": fadeexp,
            "accum": accum,
            "termthresh": termthresh,
            "griddim": griddim,
            "blocksize": blocksize,
            "bwdblocksize": bwdblocksize,
        },
    )
    return out
