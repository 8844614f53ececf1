"""Primitive placement on the mesh: the barycentric half of the decoder -> raymarch hand-off.

The reference's DecoderAssembler builds a full 1024 x 1024 position map per batch element with three index_selects
(models/decoders/assembler.py:118-122) and then reads three texels per primitive from it (assembler.py:143-206).
``prim_placement`` returns exactly what those reads produce -- ``primpos`` (before the residuals are added),
``vcenterdu`` and ``vcenterdv`` -- from one HIP kernel (9 vertex fetches per primitive), bit-identical in the forward
and differentiable with respect to ``geo``.  There is no CPU path.
"""
import torch
from torch.autograd import Function

from . import _lib
from ._tensors import ptr, require_device_f32, stream_ptr

# number of primitives -> (ny, nx, y0, sy, x0, sx): the centre grids of the two branches of the reference that also
# define vcenterdu / vcenterdv (assembler.py:143-170 and :180-206); the others raise there (:217-224)
GRIDS = {256: (16, 16, 32, 64, 32, 64), 16384: (128, 128, 4, 8, 4, 8)}


class _Placement(Function):
    @staticmethod
    def forward(ctx, geo, idxim32, barim, volradius, grid):
        B, V = geo.shape[0], geo.shape[1]
        T = idxim32.shape[0]
        ny, nx, y0, sy, x0, sx = grid
        primpos = torch.empty((B, ny * nx, 3), dtype=torch.float32, device=geo.device)
        du = torch.empty((B, ny, nx, 3), dtype=torch.float32, device=geo.device)
        dv = torch.empty((B, ny, nx, 3), dtype=torch.float32, device=geo.device)
        stream = torch.cuda.current_stream(geo.device).cuda_stream
        with torch.cuda.device(geo.device):
            _lib.check(_lib.get_lib().mvp_prim_placement_forward(
                B, V, T, ny, nx, y0, sy, x0, sx, float(volradius), geo.data_ptr(), idxim32.data_ptr(), barim.data_ptr(),
                primpos.data_ptr(), du.data_ptr(), dv.data_ptr(), stream), "mvp_prim_placement_forward")
        ctx.save_for_backward(idxim32, barim)
        ctx.meta = (B, V, T, grid, float(volradius))
        return primpos, du, dv

    @staticmethod
    def backward(ctx, g_pos, g_du, g_dv):
        idxim32, barim = ctx.saved_tensors
        B, V, T, (ny, nx, y0, sy, x0, sx), volradius = ctx.meta
        grad_geo = torch.empty((B, V, 3), dtype=torch.float32, device=idxim32.device)
        ptr = lambda t: 0 if t is None else t.contiguous().data_ptr()
        keep = [None if t is None else t.contiguous() for t in (g_pos, g_du, g_dv)]
        stream = torch.cuda.current_stream(idxim32.device).cuda_stream
        with torch.cuda.device(idxim32.device):
            _lib.check(_lib.get_lib().mvp_prim_placement_backward(
                B, V, T, ny, nx, y0, sy, x0, sx, volradius, idxim32.data_ptr(), barim.data_ptr(), ptr(keep[0]), ptr(keep[1]),
                ptr(keep[2]), grad_geo.data_ptr(), stream), "mvp_prim_placement_backward")
        return grad_geo, None, None, None, None


def _as_int32(idxim):
    """The reference registers idxim as int64 (assembler.py:63); the kernel reads int32.  Converted once per buffer:
    the copy rides on the source tensor OBJECT (an address-keyed cache would be fooled by the caching allocator)."""
    if idxim.dtype == torch.int32 and idxim.is_contiguous():
        return idxim
    hit = getattr(idxim, "_mvp_int32", None)
    if hit is None or hit[0] != idxim._version:
        hit = (idxim._version, idxim.to(torch.int32).contiguous())
        idxim._mvp_int32 = hit
    return hit[1]


def _check_indices(idxim, V):
    """The kernels index geo with idxim directly; the reference's index_select raises on an index outside [0, V)
    (assembler.py:118-122), so this does too -- once per index tensor OBJECT, version and V: two device reductions and a
    host synchronisation the first time only (the index map is a registered buffer: one tensor for the life of the
    module).  The mark rides on the tensor object, like the int32 copy: an address-keyed cache would be fooled by the
    caching allocator, which hands a freed address to the next tensor of the same size with _version 0 -- and an
    unvalidated index is an out-of-bounds device access.  A fresh tensor per call is validated per call.  Skipped while a
    stream is being captured (nothing here can be captured; a captured graph replays validated tensors)."""
    key = (idxim._version, int(V))
    if getattr(idxim, "_mvp_checked", None) == key or torch.cuda.is_current_stream_capturing():
        return
    lo, hi = int(idxim.min().item()), int(idxim.max().item())
    if lo < 0 or hi >= V:
        raise IndexError("idxim holds vertex indices in [%d, %d] but geo has %d vertices" % (lo, hi, V))
    idxim._mvp_checked = key


def prim_placement(geo, idxim, barim, volradius, nprims):
    """geo [B,V,3] float32 (de-normalised vertices), idxim [T,T,3] integer, barim [T,T,3] float32.
    Returns (primpos [B,nprims,3], vcenterdu [B,ny,nx,3], vcenterdv [B,ny,nx,3]) as assembler.py:143-206 computes them
    from its postex map."""
    if nprims not in GRIDS:
        raise ValueError("Unsupported number of primitives for mesh placement: %r (the reference defines the u/v "
                         "centres only for %s)" % (nprims, sorted(GRIDS)))
    require_device_f32("geo", geo)
    require_device_f32("barim", barim)
    if geo.dim() != 3 or geo.shape[2] != 3:
        raise RuntimeError("geo must be [B, V, 3]")
    if idxim.dim() != 3 or idxim.shape[2] != 3 or idxim.shape[0] != idxim.shape[1] or tuple(barim.shape) != tuple(idxim.shape):
        raise RuntimeError("idxim / barim must be [T, T, 3] with equal shapes")
    if not idxim.is_cuda or idxim.dtype not in (torch.int32, torch.int64):
        raise RuntimeError("idxim must be an int32/int64 tensor on the GPU")
    _check_indices(idxim, geo.shape[1])
    return _Placement.apply(geo, _as_int32(idxim), barim, float(volradius), GRIDS[nprims])


# ---- residual half of the hand-off (csrc/primpose.hip; assembler.py:241-253) ---------------------------------------------
def _frames_or_shared(name, t, N, K, tail):
    """[N, K, *tail] -> (tensor, frame stride in floats); [K, *tail] or [1, K, *tail] -> shared by the frames (stride 0)."""
    t = require_device_f32(name, t.contiguous() if torch.is_tensor(t) else t)
    per = 1
    for d in tail:
        per *= d
    if tuple(t.shape) == (K,) + tail or tuple(t.shape) == (1, K) + tail:
        return t, 0
    if tuple(t.shape) == (N, K) + tail:
        return t, K * per
    raise RuntimeError("%s must be [N, K, %s] or [K, %s] (N = %d, K = %d), got %s"
                       % (name, ", ".join(map(str, tail)), ", ".join(map(str, tail)), N, K, tuple(t.shape)))


def _scale0_strides(scale0, N, K):
    """Strides (frame, primitive, component) in floats of a base scale that broadcasts to [N, K, 3]."""
    e = scale0.expand(N, K, 3) if scale0.dim() == 3 else scale0.reshape((1,) * (3 - scale0.dim()) + tuple(scale0.shape)).expand(N, K, 3)
    return tuple(int(s) for s in e.stride())


class _PrimResiduals(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pos0, rot0, scale0, posres, rotres, scaleres, rw, N, K):
        pos0, pos0_sn = _frames_or_shared("pos0", pos0, N, K, (3,))
        rot0, rot0_sn = _frames_or_shared("rot0", rot0, N, K, (3, 3))
        posres, posres_sn = _frames_or_shared("posres", posres, N, K, (3,))
        rotres, rotres_sn = _frames_or_shared("rotres", rotres, N, K, (3,))
        scaleres, scaleres_sn = _frames_or_shared("scaleres", scaleres, N, K, (3,))
        scale0 = scale0.detach()
        if not scale0.is_cuda or scale0.dtype != torch.float32:
            raise RuntimeError("scale0 must be a float32 tensor on the GPU")
        ssn, ssk, ssc = _scale0_strides(scale0, N, K)
        dev = pos0.device
        primpos = torch.empty((N, K, 3), device=dev, dtype=torch.float32)
        primrot = torch.empty((N, K, 3, 3), device=dev, dtype=torch.float32)
        primscale = torch.empty((N, K, 3), device=dev, dtype=torch.float32)
        ins = (N, K, float(rw), ptr(pos0), pos0_sn, ptr(rot0), rot0_sn, ptr(scale0), ssn, ssk, ssc, ptr(posres), posres_sn,
               ptr(rotres), rotres_sn, ptr(scaleres), scaleres_sn)
        with torch.cuda.device(dev):
            _lib.check(_lib.get_lib().mvp_prim_residuals_forward(*ins, ptr(primpos), ptr(primrot), ptr(primscale),
                                                                 stream_ptr(dev)), "mvp_prim_residuals_forward")
        ctx.save_for_backward(pos0, rot0, scale0, posres, rotres, scaleres)
        ctx.meta = (N, K, float(rw), pos0_sn, rot0_sn, (ssn, ssk, ssc), posres_sn, rotres_sn, scaleres_sn)
        return primpos, primrot, primscale

    @staticmethod
    def backward(ctx, gpos, grot, gscale):
        pos0, rot0, scale0, posres, rotres, scaleres = ctx.saved_tensors
        N, K, rw, pos0_sn, rot0_sn, (ssn, ssk, ssc), posres_sn, rotres_sn, scaleres_sn = ctx.meta
        dev = pos0.device

        def dense(g, shape):  # an output nobody differentiated arrives as None
            return torch.zeros(shape, device=dev, dtype=torch.float32) if g is None else g.contiguous().float()

        gpos, grot, gscale = dense(gpos, (N, K, 3)), dense(grot, (N, K, 3, 3)), dense(gscale, (N, K, 3))
        need = ctx.needs_input_grad
        g_pos0 = torch.empty_like(pos0) if need[0] else None
        g_rot0 = torch.empty_like(rot0) if need[1] else None
        g_posres, g_rotres, g_scaleres = torch.empty_like(posres), torch.empty_like(rotres), torch.empty_like(scaleres)
        ins = (N, K, rw, ptr(pos0), pos0_sn, ptr(rot0), rot0_sn, ptr(scale0), ssn, ssk, ssc, ptr(posres), posres_sn,
               ptr(rotres), rotres_sn, ptr(scaleres), scaleres_sn)
        with torch.cuda.device(dev):
            _lib.check(_lib.get_lib().mvp_prim_residuals_backward(
                *ins, ptr(gpos), ptr(grot), ptr(gscale), ptr(g_pos0) if g_pos0 is not None else None,
                ptr(g_rot0) if g_rot0 is not None else None, ptr(g_posres), ptr(g_rotres), ptr(g_scaleres), stream_ptr(dev)),
                "mvp_prim_residuals_backward")
        return (g_pos0, g_rot0, None, g_posres if need[3] else None, g_rotres if need[4] else None,
                g_scaleres if need[5] else None, None, None, None)


def prim_residuals(pos0, rot0, scale0, posres, rotres, scaleres, residuals_weight, nframes):
    """The reference's residual composition (assembler.py:241-253) as one kernel each way:
        rw = clamp(residuals_weight, 0, 1); if rw < 1: posres *= rw; rotres *= rw; scaleres = scaleres * rw + (1 - rw)
        primpos = pos0 + posres;  primrot = rot0 @ rodrigues(rotres);  primscale = scale0 * scaleres
    pos0 / posres / rotres / scaleres: [N, K, 3] or [K, 3] (shared by the frames); rot0: [N, K, 3, 3] or [K, 3, 3]; scale0: a
    tensor that broadcasts to [N, K, 3] (no gradient: a buffer in the reference).  Returns primpos [N, K, 3], primrot
    [N, K, 3, 3], primscale [N, K, 3]."""
    K = posres.shape[-2]
    rw = min(max(float(residuals_weight), 0.0), 1.0)   # sorted([0, w, 1])[1]
    return _PrimResiduals.apply(pos0, rot0, scale0, posres, rotres, scaleres, rw, int(nframes), int(K))


class _PrimFrame(torch.autograd.Function):
    @staticmethod
    def forward(ctx, du, dv):
        du, dv = require_device_f32("vcenterdu", du.contiguous()), require_device_f32("vcenterdv", dv.contiguous())
        if du.shape != dv.shape or du.shape[-1] != 3:
            raise RuntimeError("vcenterdu / vcenterdv must have equal shapes [..., 3]")
        M = du.numel() // 3
        rot = torch.empty(du.shape[:-1] + (3, 3), device=du.device, dtype=torch.float32)
        with torch.cuda.device(du.device):
            _lib.check(_lib.get_lib().mvp_prim_frame_forward(M, ptr(du), ptr(dv), ptr(rot), stream_ptr(du.device)),
                       "mvp_prim_frame_forward")
        ctx.save_for_backward(du, dv)
        return rot

    @staticmethod
    def backward(ctx, g):
        du, dv = ctx.saved_tensors
        g = g.contiguous().float()
        gdu, gdv = torch.empty_like(du), torch.empty_like(dv)
        with torch.cuda.device(du.device):
            _lib.check(_lib.get_lib().mvp_prim_frame_backward(du.numel() // 3, ptr(du), ptr(dv), ptr(g), ptr(gdu), ptr(gdv),
                                                              stream_ptr(du.device)), "mvp_prim_frame_backward")
        return gdu, gdv


def prim_frame(vcenterdu, vcenterdv):
    """The TBN matrix of assembler.py:226-239 as one kernel each way: vcenterdu / vcenterdv [B, ny, nx, 3] (prim_placement's
    outputs) -> primrot [B, ny*nx, 3, 3] with columns tangent, bitangent, normal."""
    rot = _PrimFrame.apply(vcenterdu, vcenterdv)
    return rot.reshape(rot.shape[0], -1, 3, 3) if rot.dim() > 3 else rot
