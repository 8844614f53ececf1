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


class AssembleTemplateFrames(Function):
    """tplate[f] = gain[f] * assemble(tex[0], opacity[0]): the frame-broadcast form (csrc/assemble.hip, second half)."""

    @staticmethod
    def forward(ctx, tex, opacity, gain, nboxes, boxsize):
        tex = require_device_f32("tex", tex)
        opacity = require_device_f32("opacity", opacity)
        gain = require_device_f32("gain", gain)
        nh = int(math.isqrt(nboxes))
        assert nh * nh == nboxes, "nboxes must be a square (rgb.py:130-131)"
        S, F = nh * boxsize, gain.numel()
        assert tex.shape == (1, 3 * boxsize, S, S) and opacity.shape == (1, boxsize, S, S) and gain.dim() == 1
        dev = tex.device
        tplate = torch.empty((F, nboxes, boxsize, boxsize, boxsize, 4), device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(_lib.get_lib().mvp_template_assemble_frames_forward(F, nh, boxsize, ptr(tex), ptr(opacity), ptr(gain),
                                                                           ptr(tplate), stream_ptr(dev)),
                       "mvp_template_assemble_frames_forward")
        ctx.save_for_backward(tex, opacity, gain)
        ctx.dims = (F, nh, boxsize)
        return tplate

    @staticmethod
    def backward(ctx, grad_tplate):
        tex, opacity, gain = ctx.saved_tensors
        F, nh, B = ctx.dims
        dev = tex.device
        grad_tplate = grad_tplate.contiguous().float()
        gtex, gop = torch.empty_like(tex), torch.empty_like(opacity)
        lib = _lib.get_lib()
        blocks = int(lib.mvp_template_assemble_frames_blocks(nh, B))
        partials = torch.empty((max(blocks, 1), max(F, 1)), device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(lib.mvp_template_assemble_frames_backward(F, nh, B, ptr(tex), ptr(opacity), ptr(gain), ptr(grad_tplate),
                                                                 ptr(gtex), ptr(gop), ptr(partials), stream_ptr(dev)),
                       "mvp_template_assemble_frames_backward")
        ggain = partials.sum(0)[:F] if ctx.needs_input_grad[2] else None
        return gtex, gop, ggain, None, None


def assemble_template_frames(tex, opacity, gain, nboxes, boxsize=8):
    """tex [1,3*B,nh*B,nh*B], opacity [1,B,nh*B,nh*B], gain [F] -> template [F,nboxes,B,B,B,4] = gain[f] * slabs."""
    return AssembleTemplateFrames.apply(tex, opacity, gain, nboxes, boxsize)


def assemble_template(tex, opacity, nboxes, boxsize=8):
    """tex [N,3*B,nh*B,nh*B], opacity [N,B,nh*B,nh*B] -> template [N,nboxes,B,B,B,4] (channels-last slabs)."""
    return AssembleTemplate.apply(tex, opacity, nboxes, boxsize)
