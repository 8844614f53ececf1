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
