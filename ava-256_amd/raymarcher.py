"""`Raymarcher` -- the nn.Module the autoencoder instantiates (`raymarcherlib.Raymarcher(volradius)`).

API contract taken from the reference (models/raymarchers/mvpraymarcher.py:17-54): constructor
`(volradius, dt=1.0)`, attributes `volume_radius` and `dt` (= dt / volradius, the march step in volume
units), and `forward(raypos, raydir, tminmax, decout, renderoptions={}, rayterm=None, with_pos_img=None)`
returning `(rayrgb [N,3,H,W], rayalpha [N,1,H,W], rayrgba [N,4,H,W] view, None)`.  The module owns no
parameters or buffers, so `state_dict()` keys of a model that embeds it are unchanged.
"""
import inspect
from typing import Dict, Optional

import torch
from torch import nn

from . import _lib
from ._tensors import aligned, ptr, require_device_f32, stream_ptr
from .mvpraymarch import mvpraymarch, mvpraymarch_from_cameras

# renderoptions are forwarded only when they name a keyword of mvpraymarch (the reference filters with
# mvpraymarch.__code__.co_varnames, mvpraymarcher.py:45; the parameter list is the same set of names)
_OPTION_NAMES = frozenset(inspect.signature(mvpraymarch).parameters) - {
    "raypos", "raydir", "stepsize", "tminmax", "primtransf", "template", "warp", "rayterm"}


class _SplitRGBA(torch.autograd.Function):
    """[N,H,W,4] -> rgb [N,3,H,W], alpha [N,1,H,W] in one pass each way (mvp_rgba_split_*; mvpraymarcher.py:50-51)."""

    @staticmethod
    def forward(ctx, rgba):
        N, H, W = rgba.shape[0], rgba.shape[1], rgba.shape[2]
        rgb = torch.empty((N, 3, H, W), dtype=torch.float32, device=rgba.device)
        alpha = torch.empty((N, 1, H, W), dtype=torch.float32, device=rgba.device)
        with torch.cuda.device(rgba.device):
            _lib.check(_lib.get_lib().mvp_rgba_split_forward(N, H, W, ptr(rgba), ptr(rgb), ptr(alpha),
                                                             stream_ptr(rgba.device)), "mvp_rgba_split_forward")
        ctx.shape = (N, H, W)
        return rgb, alpha

    @staticmethod
    def backward(ctx, g_rgb, g_alpha):
        N, H, W = ctx.shape
        dev = (g_rgb if g_rgb is not None else g_alpha).device
        g_rgb = None if g_rgb is None else g_rgb.contiguous()
        g_alpha = None if g_alpha is None else g_alpha.contiguous()
        g = torch.empty((N, H, W, 4), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.get_lib().mvp_rgba_split_backward(N, H, W, ptr(g_rgb), ptr(g_alpha), ptr(g), stream_ptr(dev)),
                       "mvp_rgba_split_backward")
        return g


def split_rgba_nchw(rayrgba_nhwc: torch.Tensor):
    """[N,H,W,4] -> (rgb [N,3,H,W] contiguous, alpha [N,1,H,W] contiguous, rgba [N,4,H,W] view)."""
    rayrgba_nhwc = aligned(require_device_f32("rayrgba", rayrgba_nhwc))
    rgb, alpha = _SplitRGBA.apply(rayrgba_nhwc)
    return rgb, alpha, rayrgba_nhwc.movedim(3, 1)


class Raymarcher(nn.Module):
    def __init__(self, volradius, dt: float = 1.0):
        super().__init__()
        self.volume_radius = volradius
        self.dt = dt / volradius

    def forward(self, raypos: torch.Tensor, raydir: torch.Tensor, tminmax: torch.Tensor,
                decout: Dict[str, torch.Tensor], renderoptions: Optional[dict] = {}, rayterm=None,
                with_pos_img=None):
        opts = {k: v for k, v in (renderoptions or {}).items() if k in _OPTION_NAMES}
        prims = (decout["primpos"], decout["primrot"], decout["primscale"])
        rayrgba = mvpraymarch(raypos, raydir, self.dt, tminmax, prims, template=decout["template"],
                              warp=decout.get("warp", None), rayterm=rayterm, **opts)
        rayrgb, rayalpha, rayrgba_nchw = split_rgba_nchw(rayrgba)
        return rayrgb, rayalpha, rayrgba_nchw, None  # pos_img is always None in the reference as well

    def forward_from_cameras(self, campos: torch.Tensor, camrot: torch.Tensor, focal: torch.Tensor,
                             princpt: torch.Tensor, pixelcoords, decout: Dict[str, torch.Tensor],
                             renderoptions: Optional[dict] = {}):
        """The caller's two statements -- compute_raydirs(campos, camrot, focal, princpt, pixelcoords, volume_radius)
        then forward(raypos, raydir, tminmax, decout) (models/autoencoder.py:240-252) -- as one kernel pass: the rays
        are made inside the march (mvp_march_forward_cams), bit-identical to the two-call form.  Rendering: no ray
        tensor touches HBM.  Training: the forward march writes the rays once, for its backward -- no raydirs launch,
        no ray reads in the forward.  Optional extension; the drop-in path is forward()."""
        opts = {k: v for k, v in (renderoptions or {}).items() if k in _OPTION_NAMES}
        prims = (decout["primpos"], decout["primrot"], decout["primscale"])
        rayrgba = mvpraymarch_from_cameras(campos, camrot, focal, princpt, pixelcoords, self.volume_radius, self.dt,
                                           prims, decout["template"], **opts)
        rayrgb, rayalpha, rayrgba_nchw = split_rgba_nchw(rayrgba)
        return rayrgb, rayalpha, rayrgba_nchw, None
