"""CPU checker for the primitive-placement kernel (barycentric half of row N2).  TEST INFRASTRUCTURE ONLY.

Restates /root/reference/models/decoders/assembler.py:118-122 (the postex map) and the reads the assembler makes of it
for the two primitive counts whose branches define the u/v centres (assembler.py:143-170, :180-206), plus the gradient
with respect to `geo`.  The forward keeps the reference's operation order in the array dtype (float32 in -> the same
float32 roundings as the eager torch expression); the backward accumulates in float64.
Pinned by tests/golden/placement_ref.npz, produced by executing those reference lines themselves
(tests/golden/gen_placement.py).
"""
import numpy as np

GRIDS = {256: (16, 16, 32, 64, 32, 64), 16384: (128, 128, 4, 8, 4, 8)}  # ny, nx, y0, sy, x0, sx


def postex_map(geo, idxim, barim, volradius):
    """geo [B,V,3], idxim [T,T,3] int, barim [T,T,3] -> postex [B,3,T,T] (assembler.py:118-122)."""
    g0, g1, g2 = geo[:, idxim[:, :, 0]], geo[:, idxim[:, :, 1]], geo[:, idxim[:, :, 2]]  # [B,T,T,3]
    p = (barim[None, :, :, 0, None] * g0 + barim[None, :, :, 1, None] * g1) + barim[None, :, :, 2, None] * g2
    return np.transpose(p, (0, 3, 1, 2)) / np.asarray(volradius, dtype=geo.dtype)


def placement(geo, idxim, barim, volradius, nprims):
    ny, nx, y0, sy, x0, sx = GRIDS[nprims]
    post = postex_map(geo, idxim, barim, volradius)
    B = geo.shape[0]
    primpos = np.ascontiguousarray(np.transpose(post[:, :, y0::sy, x0::sx], (0, 2, 3, 1))).reshape(B, nprims, 3)
    geodu = post[:, :, :, 1:] - post[:, :, :, :-1]
    geodv = post[:, :, 1:, :] - post[:, :, :-1, :]
    du = np.transpose(geodu[:, :, y0::sy, x0::sx], (0, 2, 3, 1))
    dv = np.transpose(geodv[:, :, y0::sy, x0::sx], (0, 2, 3, 1))
    return primpos, np.ascontiguousarray(du), np.ascontiguousarray(dv)


def placement_backward(geo_shape, idxim, barim, volradius, nprims, g_pos, g_du, g_dv):
    """d(sum(primpos*g_pos + du*g_du + dv*g_dv)) / d geo, float64."""
    ny, nx, y0, sy, x0, sx = GRIDS[nprims]
    B, V = geo_shape[0], geo_shape[1]
    out = np.zeros((B, V, 3), np.float64)
    ys, xs = y0 + sy * np.arange(ny), x0 + sx * np.arange(nx)
    Y, X = np.meshgrid(ys, xs, indexing="ij")
    gp = np.asarray(g_pos, np.float64).reshape(B, ny, nx, 3)
    gu, gv = np.asarray(g_du, np.float64), np.asarray(g_dv, np.float64)
    for (yy, xx, g) in ((Y, X, gp - gu - gv), (Y, X + 1, gu), (Y + 1, X, gv)):
        for c in range(3):
            idx = idxim[yy, xx, c].reshape(-1)
            w = barim[yy, xx, c].astype(np.float64).reshape(-1, 1)
            for b in range(B):
                np.add.at(out[b], idx, w * g[b].reshape(-1, 3) / float(volradius))
    return out
