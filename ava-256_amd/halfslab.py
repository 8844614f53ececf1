"""halfslab -- the opt-in render path over half-precision slabs (round 5).

The reference's kernels read fp32 RGBA slabs (extensions/mvpraymarch/primsampler.h:44-66, utils.h:408-502; the decoder
hand-off models/decoders/assembler.py:261 makes them).  BASELINE.json's configs[1] is labelled "bf16" and the forward's
sweep is bound by its gathers, so a caller that RENDERS (no gradients) may hand the march fp16 RGBA slabs instead:

    th = template_to_half(template)                        # or assemble_template_half(tex, opacity, nboxes)
    rgba = render_half(raypos, raydir, stepsize, tminmax, (primpos, primrot, primscale), th)
    rgba = render_half_from_cameras(campos, camrot, focal, princpt, pixelcoords, volradius, stepsize, primtransf, th)

Same sample set, weights, interpolation and compositing in fp32; only the slab VALUES carry the fp16 storage rounding
(2^-11 relative).  Never the training path and never the headline number: `mvpraymarch(...)` keeps fp32 slabs, and these
functions refuse to run with gradients enabled on their inputs.  8^3 slabs (the reference's size)."""
import math

import torch

from . import _hooks, _lib
from ._tensors import aligned, ptr, require_device_f32, stream_ptr
from .mvpraymarch import build_accel


def template_to_half(template):
    """[N,K,TD,TH,TW,4] float32 -> float16, round to nearest even (one pass: 16 B read + 8 B written per voxel)."""
    template = aligned(require_device_f32("template", template))
    assert template.dim() == 6 and template.size(-1) == 4
    out = torch.empty(template.shape, device=template.device, dtype=torch.float16)
    with torch.cuda.device(template.device), _hooks.timed("template_to_half", template.device):
        _lib.check(_lib.get_lib().mvp_template_to_half(template.numel() // 4, ptr(template), ptr(out),
                                                       stream_ptr(template.device)), "mvp_template_to_half")
    return out


def assemble_template_half(tex, opacity, nboxes, boxsize=8):
    """assemble_template(...) writing fp16 slabs directly: tex [N,3*B,nh*B,nh*B], opacity [N,B,nh*B,nh*B] ->
    [N,nboxes,B,B,B,4] float16 (the decoder hand-off of rgb.py:137-143 / geometry.py:183-185 / assembler.py:261)."""
    tex = require_device_f32("tex", tex)
    opacity = require_device_f32("opacity", opacity)
    nh = int(math.isqrt(nboxes))
    assert nh * nh == nboxes, "nboxes must be a square (rgb.py:130-131)"
    N, S = tex.size(0), nh * boxsize
    assert tex.shape == (N, 3 * boxsize, S, S) and opacity.shape == (N, boxsize, S, S)
    out = torch.empty((N, nboxes, boxsize, boxsize, boxsize, 4), device=tex.device, dtype=torch.float16)
    with torch.cuda.device(tex.device), _hooks.timed("assemble_half", tex.device):
        _lib.check(_lib.get_lib().mvp_template_assemble_forward_half(N, nh, boxsize, ptr(tex), ptr(opacity), ptr(out),
                                                                     stream_ptr(tex.device)),
                   "mvp_template_assemble_forward_half")
    return out


def _prims(primtransf):
    if isinstance(primtransf, tuple):
        primpos, primrot, primscale = primtransf
    else:  # packed [N,K,5,3] (mvpraymarch.py:355-360)
        primpos, primrot, primscale = (primtransf[:, :, 0, :].contiguous(), primtransf[:, :, 1:4, :].contiguous(),
                                       primtransf[:, :, 4, :].contiguous())
    return (require_device_f32("primpos", primpos), require_device_f32("primrot", primrot),
            require_device_f32("primscale", primscale))


def _check_half(template_half, N, K):
    if not torch.is_tensor(template_half) or template_half.dtype != torch.float16 or not template_half.is_cuda:
        raise RuntimeError("template_half must be a float16 device tensor (template_to_half / assemble_template_half)")
    if not template_half.is_contiguous():
        raise RuntimeError("template_half must be contiguous")
    assert template_half.dim() == 6 and template_half.size(-1) == 4 and template_half.shape[:2] == (N, K)
    if tuple(template_half.shape[2:5]) != (8, 8, 8):
        raise NotImplementedError("the half-precision render path takes 8^3 slabs")
    return aligned(template_half)


def _check_prims(primpos, primrot, primscale, N):
    """The shape contract of the fp32 operator (mvpraymarch.py: _forward_impl; reference mvpraymarch.py:112-127): one
    primitive set PER IMAGE.  build_accel sizes the node boxes from primpos.size(0), so a [1,K,..] avatar rendered from N > 1
    cameras would make the kernel read boxes and poses of images that do not exist."""
    K = primpos.size(1) if primpos.dim() == 3 else -1
    assert primpos.shape == (N, K, 3) and primrot.shape == (N, K, 3, 3) and primscale.shape == (N, K, 3), \
        "primpos / primrot / primscale must be [N,K,3] / [N,K,3,3] / [N,K,3] with N = %d images" % N
    return K


def _same_device(*tensors):
    devs = {t.device for t in tensors if torch.is_tensor(t)}
    if len(devs) > 1:
        raise RuntimeError("all inputs must live on one device, got %s" % sorted(str(d) for d in devs))


def _no_grad(*tensors):
    if torch.is_grad_enabled() and any(torch.is_tensor(t) and t.requires_grad for t in tensors):
        raise RuntimeError("the half-precision slab path renders only: call it under torch.no_grad() (training keeps fp32 "
                           "slabs: mvpraymarch)")


def render_half(raypos, raydir, stepsize, tminmax, primtransf, template_half, fadescale=8.0, fadeexp=8.0):
    """mvpraymarch(...) without gradients over fp16 slabs -> rayrgba [N,H,W,4] float32."""
    primpos, primrot, primscale = _prims(primtransf)
    _no_grad(raypos, raydir, tminmax, primpos, primrot, primscale, template_half)
    raypos = aligned(require_device_f32("raypos", raypos))
    raydir = aligned(require_device_f32("raydir", raydir))
    tminmax = aligned(require_device_f32("tminmax", tminmax))
    assert raypos.dim() == 4 and raypos.size(3) == 3
    N, H, W = raypos.size(0), raypos.size(1), raypos.size(2)
    assert raydir.shape == raypos.shape and tminmax.shape == raypos.shape[:3] + (2,)
    K = _check_prims(primpos, primrot, primscale, N)
    th = _check_half(template_half, N, K)
    _same_device(raypos, raydir, tminmax, primpos, primrot, primscale, th)
    dev = primpos.device
    with torch.no_grad():
        _, _, nodeaabb = build_accel((primpos, primrot, primscale), 0, fixedorder=True)
        rayrgba = torch.empty((N, H, W, 4), device=dev, dtype=torch.float32)
        with torch.cuda.device(dev), _hooks.timed("march_render_half", dev):
            _lib.check(_lib.get_lib().mvp_march_render_half(
                N, H, W, K, ptr(raypos), ptr(raydir), ptr(tminmax), None, None, None, None, None, 1.0, float(stepsize),
                ptr(nodeaabb), ptr(primpos), ptr(primrot), ptr(primscale), 8, 8, 8, ptr(th), ptr(rayrgba),
                float(fadescale), float(fadeexp), ptr(_hooks.diag), stream_ptr(dev)), "mvp_march_render_half")
    return rayrgba


def render_half_from_cameras(campos, camrot, focal, princpt, pixelcoords, volradius, stepsize, primtransf, template_half,
                             fadescale=8.0, fadeexp=8.0):
    """mvpraymarch_from_cameras(...) without gradients over fp16 slabs (rays made inside the march)."""
    primpos, primrot, primscale = _prims(primtransf)
    _no_grad(campos, camrot, focal, princpt, primpos, primrot, primscale, template_half)
    campos, camrot = require_device_f32("campos", campos), require_device_f32("camrot", camrot)
    focal, princpt = require_device_f32("focal", focal), require_device_f32("princpt", princpt)
    N = campos.size(0)
    assert campos.shape == (N, 3) and camrot.shape == (N, 3, 3) and focal.shape == (N, 2) and princpt.shape == (N, 2)
    if isinstance(pixelcoords, tuple):
        W, H = pixelcoords
        pc = None
    else:
        pc = aligned(require_device_f32("pixelcoords", pixelcoords))
        assert pc.dim() == 4 and pc.size(0) == N and pc.size(3) == 2
        H, W = pc.size(1), pc.size(2)
    K = _check_prims(primpos, primrot, primscale, N)
    th = _check_half(template_half, N, K)
    _same_device(campos, camrot, focal, princpt, pc, primpos, primrot, primscale, th)
    dev = primpos.device
    with torch.no_grad():
        _, _, nodeaabb = build_accel((primpos, primrot, primscale), 0, fixedorder=True)
        rayrgba = torch.empty((N, H, W, 4), device=dev, dtype=torch.float32)
        with torch.cuda.device(dev), _hooks.timed("march_render_half", dev):
            _lib.check(_lib.get_lib().mvp_march_render_half(
                N, H, W, K, None, None, None, ptr(campos), ptr(camrot), ptr(focal), ptr(princpt), ptr(pc),
                float(volradius), float(stepsize), ptr(nodeaabb), ptr(primpos), ptr(primrot), ptr(primscale), 8, 8, 8,
                ptr(th), ptr(rayrgba), float(fadescale), float(fadeexp), ptr(_hooks.diag), stream_ptr(dev)),
                "mvp_march_render_half")
    return rayrgba
