"""assemble_template -- decoder -> raymarch hand-off in one pass.

Replaces, for a caller that is willing to hand over the raw decoder outputs, the three eager steps of the reference:
`RGBDecoder`'s view/permute/reshape (models/decoders/rgb.py:137-143), `GeometryDecoder`'s (geometry.py:183-185) and
`template = cat([relu(primrgb * 25 + 100), relu(primalpha)], -1)` (models/decoders/assembler.py:261).
"""
import math

import torch
from torch.autograd import Function

from . import _lib
from ._tensors import ptr, require_device_f32, stream_ptr


class AssembleTemplate(Function):
    @staticmethod
    def forward(ctx, tex, opacity, nboxes, boxsize):
        tex = require_device_f32("tex", tex)
        opacity = require_device_f32("opacity", opacity)
        nh = int(math.isqrt(nboxes))
        assert nh * nh == nboxes, "nboxes must be a square (rgb.py:130-131)"
        N, S = tex.size(0), nh * boxsize
        assert tex.shape == (N, 3 * boxsize, S, S) and opacity.shape == (N, boxsize, S, S)
        dev = tex.device
        tplate = torch.empty((N, nboxes, boxsize, boxsize, boxsize, 4), device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(_lib.get_lib().mvp_template_assemble_forward(N, nh, boxsize, ptr(tex), ptr(opacity), ptr(tplate),
                                                                    stream_ptr(dev)), "mvp_template_assemble_forward")
        ctx.save_for_backward(tplate)
        ctx.dims = (N, nh, boxsize)
        return tplate

    @staticmethod
    def backward(ctx, grad_tplate):
        (tplate,) = ctx.saved_tensors
        N, nh, B = ctx.dims
        S = nh * B
        dev = tplate.device
        grad_tplate = grad_tplate.contiguous().float()
        gtex = torch.empty((N, 3 * B, S, S), device=dev, dtype=torch.float32)
        gop = torch.empty((N, B, S, S), device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(_lib.get_lib().mvp_template_assemble_backward(N, nh, B, ptr(tplate), ptr(grad_tplate), ptr(gtex),
                                                                     ptr(gop), stream_ptr(dev)),
                       "mvp_template_assemble_backward")
        return gtex, gop, None, None


def assemble_template(tex, opacity, nboxes, boxsize=8):
    """tex [N,3*B,nh*B,nh*B], opacity [N,B,nh*B,nh*B] -> template [N,nboxes,B,B,B,4] (channels-last slabs)."""
    return AssembleTemplate.apply(tex, opacity, nboxes, boxsize)
