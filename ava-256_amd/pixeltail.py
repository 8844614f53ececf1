"""decode_tail -- colour calibration + background matting + L1 image loss as one kernel pass each way.

The tail of the reference's `Autoencoder.decode` (models/autoencoder.py:254-265: `rayrgb = colorcal(rayrgb, cam, id)`,
`rayrgb = rayrgb + (1 - rayalpha) * bg`), its `Colorcal.forward` (models/colorcals/colorcal.py:28-31: `w * image + b`)
and the image term of the loss (losses.py:12-14 `mean_ell_1`, ddp-train.py:404-405), taken straight from the march's own
output layout `rayrgba [N,H,W,4]` (the Raymarcher's permute + two copies, mvpraymarcher.py:50-51, disappear) and handing
the march its upstream gradient in that layout as well.  `irgbrec` is bit-identical to the eager statements (same
operations, same order, one rounding each).  There is no CPU path.
"""
import torch
from torch.autograd import Function

from . import _lib
from ._tensors import aligned, ptr, require_device_f32, stream_ptr


class _DecodeTail(Function):
    @staticmethod
    def forward(ctx, rayrgba, cw, cb, bg, target):
        N, H, W = rayrgba.shape[0], rayrgba.shape[1], rayrgba.shape[2]
        dev = rayrgba.device
        lib = _lib.get_lib()
        irgbrec = torch.empty((N, 3, H, W), dtype=torch.float32, device=dev)
        ialpha = torch.empty((N, 1, H, W), dtype=torch.float32, device=dev)
        nblk = int(lib.mvp_pixel_tail_blocks(H, W))
        part = torch.empty((N, nblk), dtype=torch.float32, device=dev) if target is not None else None
        with torch.cuda.device(dev):
            _lib.check(lib.mvp_pixel_tail_forward(N, H, W, ptr(rayrgba), ptr(cw), ptr(cb), ptr(bg), ptr(target), ptr(irgbrec),
                                                  ptr(ialpha), ptr(part), stream_ptr(dev)), "mvp_pixel_tail_forward")
        ctx.save_for_backward(rayrgba, cw, bg, target, irgbrec)
        ctx.dims = (N, H, W, nblk)
        l1sum = part.sum() if part is not None else torch.zeros((), dtype=torch.float32, device=dev)
        return irgbrec, ialpha, l1sum

    @staticmethod
    def backward(ctx, g_irgbrec, g_ialpha, g_l1):
        rayrgba, cw, bg, target, irgbrec = ctx.saved_tensors
        N, H, W, nblk = ctx.dims
        dev = rayrgba.device
        g_irgbrec = None if g_irgbrec is None else g_irgbrec.contiguous().float()
        g_ialpha = None if g_ialpha is None else g_ialpha.contiguous().float()
        g_l1 = None if (g_l1 is None or target is None) else g_l1.reshape(1).contiguous().float()
        grad_rgba = torch.empty((N, H, W, 4), dtype=torch.float32, device=dev)
        grad_bg = torch.empty_like(bg) if bg is not None else None
        part = torch.empty((N, nblk, 6), dtype=torch.float32, device=dev) if cw is not None else None
        with torch.cuda.device(dev):
            _lib.check(_lib.get_lib().mvp_pixel_tail_backward(
                N, H, W, ptr(rayrgba), ptr(cw), ptr(bg), ptr(target), ptr(irgbrec), ptr(g_irgbrec), ptr(g_ialpha), ptr(g_l1),
                ptr(grad_rgba), ptr(grad_bg), ptr(part), stream_ptr(dev)), "mvp_pixel_tail_backward")
        g_cw = g_cb = None
        if part is not None:
            s = part.sum(1)
            g_cw, g_cb = s[:, :3], s[:, 3:]
        return grad_rgba, g_cw, g_cb, grad_bg, None


def decode_tail(rayrgba, cw=None, cb=None, bg=None, target=None):
    """rayrgba [N,H,W,4] (the march's output), cw / cb [N,3] (per-image colour affine, both or neither), bg [N,3,H,W] or
    None, target [N,3,H,W] or None.  Returns (irgbrec [N,3,H,W], ialpha [N,1,H,W], l1sum = sum |irgbrec - target|, a
    0-dim tensor; zero without a target).  Differentiable in rayrgba, cw, cb and bg."""
    rayrgba = aligned(require_device_f32("rayrgba", rayrgba))
    if rayrgba.dim() != 4 or rayrgba.shape[3] != 4:
        raise RuntimeError("rayrgba must be [N, H, W, 4]")
    N, H, W = rayrgba.shape[:3]
    if (cw is None) != (cb is None):
        raise RuntimeError("cw and cb come together")
    if cw is not None:
        cw, cb = require_device_f32("cw", cw.contiguous()), require_device_f32("cb", cb.contiguous())
        if cw.shape != (N, 3) or cb.shape != (N, 3):
            raise RuntimeError("cw / cb must be [N, 3]")
    for name, t in (("bg", bg), ("target", target)):
        if t is not None:
            require_device_f32(name, t)
            if t.shape != (N, 3, H, W):
                raise RuntimeError("%s must be [N, 3, H, W]" % name)
    return _DecodeTail.apply(rayrgba, cw, cb, bg, target)
