"""tests/golden/gen_primpose.py -- golden vectors of the residual composition made from the REFERENCE'S OWN LINES: the text of
/root/reference/models/decoders/assembler.py is read at run time and the statements from `rw = sorted(...)` to `primscale =
primscale * primitives_scale_residuals` (assembler.py:241-253) are executed as they stand, with a stand-in `self` whose
`rodrig` is the reference's own `Rodrigues` module (models/utils.py:470-494, imported from the mounted reference); autograd
gives the gradients.  Nothing of the reference is stored.  float64 inputs rounded to float32 values (so that the fp32 kernel
and the f64 oracle start from the same numbers), per-frame and shared residuals, residuals_weight below, at and above 1.
Run in the build container; writes tests/golden/primpose.npz."""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

if __name__ == "__main__":
    assert os.path.isdir(REF), "the reference is only mounted in the build container"
    sys.path.insert(0, REF)
    from models.utils import Rodrigues                                  # utils.py:470-494
    rodrig = Rodrigues()
    src = open(os.path.join(REF, "models/decoders/assembler.py")).read().split("\n")
    i0 = next(i for i, l in enumerate(src) if l.strip().startswith("rw = sorted([0.0, residuals_weight, 1.0])[1]"))
    i1 = next(i for i in range(i0, len(src)) if src[i].strip() == "primscale = primscale * primitives_scale_residuals")
    CODE = "\n".join(l[8:] for l in src[i0:i1 + 1])
    assert i1 - i0 == 11, (i0, i1)
    g = torch.Generator().manual_seed(41)
    out = {}
    for tag, N, K, shared, rwt in (("a", 3, 37, False, 0.35), ("b", 5, 70, True, 1.0), ("c", 2, 64, True, 1.7), ("d", 1, 5, False, 0.0)):
        def rnd(*shape, scale=1.0):
            return (scale * torch.randn(*shape, generator=g)).float().double()
        fs = (K,) if shared else (N, K)
        pos0 = rnd(N, K, 3).requires_grad_(True)
        q, _ = torch.linalg.qr(rnd(*fs, 3, 3))
        rot0 = q.float().double().requires_grad_(True)
        scale0 = (rnd(K, 1).abs() + 0.5)                                 # adaptwarps * 0.8 [K] (a buffer: no gradient)
        posres, scaleres = rnd(*fs, 3, scale=0.05).requires_grad_(True), (1.0 + rnd(*fs, 3, scale=0.1)).requires_grad_(True)
        rotres = rnd(*fs, 3, scale=0.4)
        rotres[..., :2, :] = 0.0                                         # zero rotation: theta = sqrt(1e-5)
        rotres = rotres.requires_grad_(True)
        env = {"torch": torch, "residuals_weight": rwt, "nprims": K, "self": types.SimpleNamespace(rodrig=rodrig),
               "expr_encoding": types.SimpleNamespace(size=lambda i: N),
               "primitives_position_residuals": posres.expand(N, K, 3).contiguous(),
               "primitives_rotation_residuals": rotres.expand(N, K, 3).contiguous(),
               "primitives_scale_residuals": scaleres.expand(N, K, 3).contiguous(), "primpos": pos0,
               "primrot": rot0.expand(N, K, 3, 3).contiguous(), "primscale": scale0}
        exec(CODE, env)                                                  # assembler.py:241-253, as they stand
        primpos, primrot, primscale = env["primpos"], env["primrot"], env["primscale"].expand(N, K, 3)
        gp, gr, gs = rnd(N, K, 3), rnd(N, K, 3, 3), rnd(N, K, 3)
        ((gp * primpos).sum() + (gr * primrot).sum() + (gs * primscale).sum()).backward()
        for name, t in (("pos0", pos0), ("rot0", rot0), ("scale0", scale0), ("posres", posres), ("rotres", rotres),
                        ("scaleres", scaleres), ("primpos", primpos), ("primrot", primrot), ("primscale", primscale),
                        ("g_primpos", gp), ("g_primrot", gr), ("g_primscale", gs), ("g_pos0", pos0.grad), ("g_rot0", rot0.grad),
                        ("g_posres", posres.grad), ("g_rotres", rotres.grad), ("g_scaleres", scaleres.grad)):
            out["%s_%s" % (tag, name)] = t.detach().numpy()
        out["%s_rw" % tag] = np.float64(rwt)
    np.savez_compressed(os.path.join(OUT, "primpose.npz"), **out)
    print("wrote primpose.npz:", len(out), "arrays")
