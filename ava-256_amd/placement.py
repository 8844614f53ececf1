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
from ._tensors import require_device_f32

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
