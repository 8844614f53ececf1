"""CPU checker for the fused decode tail (TEST INFRASTRUCTURE ONLY; the product never imports this).

numpy restatement of models/autoencoder.py:254-265 (colour calibration, matting), models/colorcals/colorcal.py:28-31
(`w * image + b` with w = wcam[cam] + wident[id], b = bcam[cam] + bident[id]) and losses.py:12-14 (`mean_ell_1`), with the
gradients written out by hand.  Pinned against the reference's own `Colorcal` module and `mean_ell_1` by
tests/golden/pixeltail.npz (tests/golden/gen_pixeltail.py imports them in the build container)."""
import numpy as np


def decode_tail(rayrgb, rayalpha, w, b, bg, target):
    """rayrgb [N,3,H,W], rayalpha [N,1,H,W], w / b [N,3] or None, bg / target [N,3,H,W] or None -> (irgbrec, l1 mean)."""
    out = rayrgb
    if w is not None:
        out = w[:, :, None, None] * out + b[:, :, None, None]      # colorcal.py:31
    if bg is not None:
        out = out + (1.0 - rayalpha) * bg                           # autoencoder.py:264
    l1 = None if target is None else np.abs(out - target).mean()    # losses.py:13-14
    return out, l1


def decode_tail_backward(rayrgb, rayalpha, w, bg, target, irgbrec, g_irgbrec, g_l1mean):
    """Gradients of sum(g_irgbrec * irgbrec) + g_l1mean * mean|irgbrec - target| w.r.t. rayrgb, rayalpha, w, b, bg."""
    G = np.zeros_like(irgbrec) if g_irgbrec is None else g_irgbrec.copy()
    if target is not None:
        G = G + g_l1mean * np.sign(irgbrec - target) / irgbrec.size
    g_alpha = np.zeros_like(rayalpha)
    g_bg = None
    if bg is not None:
        g_alpha = -(G * bg).sum(1, keepdims=True)
        g_bg = G * (1.0 - rayalpha)
    if w is not None:
        g_rgb = G * w[:, :, None, None]
        g_w, g_b = (G * rayrgb).sum((2, 3)), G.sum((2, 3))
    else:
        g_rgb, g_w, g_b = G, None, None
    return g_rgb, g_alpha, g_w, g_b, g_bg
