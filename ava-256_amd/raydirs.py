"""compute_raydirs -- pixel -> (raypos, raydir, tminmax), same Python surface as the reference's
extensions/utils/utils.py:21-51 (ComputeRaydirs / compute_raydirs), running mvp_raydirs_forward
(csrc/raydirs.hip) instead of utilslib.compute_raydirs_forward.
"""
import torch
from torch.autograd import Function

from . import _lib
from ._tensors import aligned, ptr, require_device_f32, stream_ptr


class ComputeRaydirs(Function):
    @staticmethod
    def forward(ctx, viewpos, viewrot, focal, princpt, pixelcoords, volradius):
        viewpos = require_device_f32("viewpos", viewpos)
        viewrot = require_device_f32("viewrot", viewrot)
        focal = require_device_f32("focal", focal)
        princpt = require_device_f32("princpt", princpt)
        N = viewpos.size(0)
        if isinstance(pixelcoords, tuple):  # (W, H): integer pixel grid (utils.py:28-30)
            W, H = pixelcoords
            pixelcoords = None
        else:
            pixelcoords = require_device_f32("pixelcoords", pixelcoords)
            H, W = pixelcoords.size(1), pixelcoords.size(2)
            assert pixelcoords.size(0) == N and pixelcoords.size(3) == 2
        assert viewpos.shape == (N, 3) and viewrot.shape == (N, 3, 3) and focal.shape == (N, 2) and princpt.shape == (N, 2)
        dev = viewpos.device
        raypos = torch.empty((N, H, W, 3), device=dev, dtype=torch.float32)
        raydirs = torch.empty((N, H, W, 3), device=dev, dtype=torch.float32)
        tminmax = torch.empty((N, H, W, 2), device=dev, dtype=torch.float32)
        pc = aligned(pixelcoords)
        with torch.cuda.device(dev):
            _lib.check(_lib.get_lib().mvp_raydirs_forward(
                N, H, W, ptr(viewpos), ptr(viewrot), ptr(focal), ptr(princpt), ptr(pc), float(volradius),
                ptr(raypos), ptr(raydirs), ptr(tminmax), stream_ptr(dev)), "mvp_raydirs_forward")
        ctx.mark_non_differentiable(raypos, raydirs, tminmax)
        return raypos, raydirs, tminmax

    @staticmethod
    def backward(ctx, grad_raypos, grad_raydirs, grad_tminmax):
        # the reference defines no gradient for ray generation (utils.py:44-46, utils_kernel.cu:54-95)
        return None, None, None, None, None, None


def compute_raydirs(viewpos, viewrot, focal, princpt, pixelcoords, volradius):
    raypos, raydirs, tminmax = ComputeRaydirs.apply(viewpos, viewrot, focal, princpt, pixelcoords, volradius)
    return raypos, raydirs, tminmax
