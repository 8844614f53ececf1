"""CPU restatement (numpy) of the decoder -> raymarch hand-off.  TEST INFRASTRUCTURE ONLY (see oracle/mvp_oracle.c).

Follows models/decoders/rgb.py:137-143 (view [N,B,3,h,B,w,B], permute (0,3,5,1,4,6,2), reshape), the identical
one-channel sequence of models/decoders/geometry.py:183-185, and models/decoders/assembler.py:261.
Pinned by tests/golden/assemble_map.npz: source/destination index pairs read off the REAL RGBDecoder / GeometryDecoder
modules of the reference run on CPU with index-encoded activations (tests/golden/gen_golden.py).
"""
import numpy as np


def src_index_rgb(nh, B, k, z, y, x, c):
    """flat index into tex [3B, S, S] (one image) of output element (k, z, y, x, c), c in 0..2"""
    S = nh * B
    hy, wx = k // nh, k % nh
    return ((z * 3 + c) * S + (hy * B + y)) * S + (wx * B + x)


def src_index_opacity(nh, B, k, z, y, x):
    S = nh * B
    hy, wx = k // nh, k % nh
    return (z * S + (hy * B + y)) * S + (wx * B + x)


def assemble_template(tex, opacity, nboxes, B=8):
    tex, opacity = np.asarray(tex), np.asarray(opacity)
    N = tex.shape[0]
    nh = int(round(np.sqrt(nboxes)))
    rgb = tex.reshape(N, B, 3, nh, B, nh, B).transpose(0, 3, 5, 1, 4, 6, 2).reshape(N, nboxes, B, B, B, 3)
    a = opacity.reshape(N, B, 1, nh, B, nh, B).transpose(0, 3, 5, 1, 4, 6, 2).reshape(N, nboxes, B, B, B, 1)
    f = tex.dtype.type
    rgb = np.maximum(rgb * f(25.0) + f(100.0), f(0.0))   # two roundings, like the eager expression
    return np.concatenate([rgb, np.maximum(a, f(0.0))], axis=-1)


def assemble_template_backward(tex, opacity, nboxes, B, g):
    """Gradients of sum(g * assemble_template(tex, opacity)) w.r.t. tex / opacity: the relu masks, the factor 25 and the
    inverse of the permute (autograd of the statements above, written out)."""
    tex, opacity, g = np.asarray(tex), np.asarray(opacity), np.asarray(g)
    N = tex.shape[0]
    nh = int(round(np.sqrt(nboxes)))
    out = assemble_template(tex, opacity, nboxes, B)
    m = (out > 0).astype(g.dtype) * g
    grgb = (m[..., :3] * g.dtype.type(25.0)).reshape(N, nh, nh, B, B, B, 3).transpose(0, 3, 6, 1, 4, 2, 5)   # [N,B,3,h,B,w,B]
    ga = m[..., 3:].reshape(N, nh, nh, B, B, B, 1).transpose(0, 3, 6, 1, 4, 2, 5)
    S = nh * B
    return grgb.reshape(N, 3 * B, S, S), ga.reshape(N, B, S, S)


def assemble_template_frames(tex, opacity, gain, nboxes, B=8):
    """Frame-broadcast form (csrc/assemble.hip, second half): tplate[f] = gain[f] * assemble(tex[0:1], opacity[0:1])."""
    base = assemble_template(tex, opacity, nboxes, B)[0]
    return np.asarray(gain).reshape(-1, 1, 1, 1, 1, 1) * base[None]


def assemble_template_frames_backward(tex, opacity, gain, nboxes, B, g):
    """-> (grad_tex, grad_opacity, grad_gain) of sum(g * assemble_template_frames(...))."""
    gain, g = np.asarray(gain), np.asarray(g)
    base = assemble_template(tex, opacity, nboxes, B)[0]
    gbase = np.tensordot(gain, g, axes=(0, 0))[None]                     # sum_f gain[f] * g[f]
    gtex, gop = assemble_template_backward(tex, opacity, nboxes, B, gbase)
    return gtex, gop, (g * base[None]).reshape(g.shape[0], -1).sum(1)
