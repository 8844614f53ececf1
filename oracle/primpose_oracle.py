"""CPU checker for the residual composition of the hand-off (TEST INFRASTRUCTURE ONLY; the product never imports this).

numpy restatement of models/decoders/assembler.py:241-253 (residual weight, position / rotation / scale residuals) and of
models/utils.py:476-494 (`Rodrigues.forward`), with the gradients written out by hand.  Pinned against the reference's own
`Rodrigues` module and autograd by tests/golden/primpose.npz (tests/golden/gen_primpose.py imports it in the build
container)."""
import numpy as np


def rodrigues(v):
    """models/utils.py:476-494 for v [..., 3] -> [..., 3, 3]."""
    theta = np.sqrt(1e-5 + (v * v).sum(-1))                                    # utils.py:477
    a = v / theta[..., None]                                                   # utils.py:478
    c, s = np.cos(theta), np.sin(theta)
    x, y, z = a[..., 0], a[..., 1], a[..., 2]
    R = np.stack([x * x + (1 - x * x) * c, x * y * (1 - c) - z * s, x * z * (1 - c) + y * s,      # utils.py:483-485
                  x * y * (1 - c) + z * s, y * y + (1 - y * y) * c, y * z * (1 - c) - x * s,      # utils.py:486-488
                  x * z * (1 - c) - y * s, y * z * (1 - c) + x * s, z * z + (1 - z * z) * c], -1)  # utils.py:489-491
    return R.reshape(v.shape[:-1] + (3, 3))


def clamp_rw(residuals_weight):
    return sorted([0.0, float(residuals_weight), 1.0])[1]                      # assembler.py:241


def prim_residuals(pos0, rot0, scale0, posres, rotres, scaleres, residuals_weight):
    """All arrays broadcast against [N, K, 3] / [N, K, 3, 3]; returns (primpos, primrot, primscale)."""
    rw = clamp_rw(residuals_weight)
    if rw < 1.0:                                                               # assembler.py:242-245
        posres, rotres, scaleres = posres * rw, rotres * rw, scaleres * rw + (1 - rw)
    primpos = pos0 + posres                                                    # assembler.py:247
    primrot = np.matmul(rot0, rodrigues(rotres))                               # assembler.py:248-251
    primscale = scale0 * scaleres                                              # assembler.py:252
    return primpos, primrot, primscale


def _unbroadcast(g, shape):
    """Sum a gradient of the broadcast shape back to the input's own shape."""
    while g.ndim > len(shape):
        g = g.sum(0)
    for i, d in enumerate(shape):
        if d == 1 and g.shape[i] != 1:
            g = g.sum(i, keepdims=True)
    return g


def prim_residuals_backward(pos0, rot0, scale0, posres, rotres, scaleres, residuals_weight, g_pos, g_rot, g_scale):
    """Gradients of sum(g_pos * primpos) + sum(g_rot * primrot) + sum(g_scale * primscale) w.r.t.
    (pos0, rot0, posres, rotres, scaleres), each in its input's own shape."""
    rw = clamp_rw(residuals_weight)
    m = rw if rw < 1.0 else 1.0
    v = rotres * m
    theta = np.sqrt(1e-5 + (v * v).sum(-1))
    a = v / theta[..., None]
    c, s = np.cos(theta)[..., None], np.sin(theta)[..., None]
    R = rodrigues(v)
    G = np.matmul(np.swapaxes(np.broadcast_to(rot0, g_rot.shape), -1, -2), g_rot)      # d/dR of <g_rot, rot0 R>
    g_rot0 = np.matmul(g_rot, np.swapaxes(np.broadcast_to(R, g_rot.shape), -1, -2))
    a_b = np.broadcast_to(a, G.shape[:-1])
    Ga = np.einsum("...ij,...j->...i", G, a_b)
    GTa = np.einsum("...ji,...j->...i", G, a_b)
    w = np.stack([G[..., 2, 1] - G[..., 1, 2], G[..., 0, 2] - G[..., 2, 0], G[..., 1, 0] - G[..., 0, 1]], -1)
    dLdc = (np.trace(G, axis1=-2, axis2=-1) - (a_b * Ga).sum(-1))[..., None]
    dLds = (a_b * w).sum(-1, keepdims=True)
    dLda = (1 - c) * (Ga + GTa) + s * w
    th = np.broadcast_to(theta[..., None], dLdc.shape)
    dLdth = -s * dLdc + c * dLds - (dLda * a_b).sum(-1, keepdims=True) / th
    g_v = dLda / th + dLdth * a_b
    sr_shape, pr_shape, rr_shape = scaleres.shape, posres.shape, rotres.shape
    return (_unbroadcast(g_pos, pos0.shape), _unbroadcast(g_rot0, rot0.shape), _unbroadcast(g_pos * m, pr_shape),
            _unbroadcast(g_v * m, rr_shape), _unbroadcast(g_scale * scale0 * m, sr_shape))


# ---- the TBN frame (models/decoders/assembler.py:227-239) ------------------------------------------------------------
def _unit(x):
    m = np.maximum(np.sqrt((x * x).sum(-1, keepdims=True)), 1e-8)      # torch.norm(...).clamp(min=1e-8)
    return x / m, m


def _unit_bwd(y, m, g):
    free = m > 1e-8
    return np.where(free, (g - y * (y * g).sum(-1, keepdims=True)) / m, g / m)


def prim_frame(du, dv):
    """vcenterdu / vcenterdv [..., 3] -> primrot [..., 3, 3] with COLUMNS tangent, bitangent, normal."""
    t, _ = _unit(du)                                                   # assembler.py:228-229
    n, _ = _unit(np.cross(t, dv))                                      # assembler.py:230-231
    b, _ = _unit(np.cross(n, t))                                       # assembler.py:232-233
    return np.stack([t, b, n], -2).swapaxes(-1, -2)                    # assembler.py:234-240 (stack rows, permute)


def prim_frame_backward(du, dv, g_rot):
    t, mt = _unit(du)
    n, mn = _unit(np.cross(t, dv))
    b, mb = _unit(np.cross(n, t))
    gt, gb, gn = g_rot[..., :, 0], g_rot[..., :, 1], g_rot[..., :, 2]
    gb0 = _unit_bwd(b, mb, gb)
    gn = gn + np.cross(t, gb0)
    gt = gt + np.cross(gb0, n)
    gn0 = _unit_bwd(n, mn, gn)
    gt = gt + np.cross(dv, gn0)
    return _unit_bwd(t, mt, gt), np.cross(gn0, t)
