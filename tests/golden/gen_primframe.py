#!/usr/bin/env python3
"""Generate tests/golden/primframe.npz from the REFERENCE'S OWN LINES (build container only).

The text of /root/reference/models/decoders/assembler.py is read at run time and the statements between `# Compute TBN
matrix` and the end of the `primrot = (...)` expression (assembler.py:227-240) are executed as they stand, on seeded float64
`vcenterdu / vcenterdv [B, n, n, 3]` (values rounded to float32 first) with a stand-in `expr_encoding`; autograd gives the
gradients of a seeded weighted sum.  Nothing of the reference is stored: the fixture holds inputs and outputs only.  Cases: a
generic grid, and one with a zero tangent difference and a dv parallel to du (the 1e-8 clamps)."""
import os
import types

import numpy as np
import torch

REF = "/root/reference/models/decoders/assembler.py"
HERE = os.path.dirname(os.path.abspath(__file__))


def reference_statements():
    src = open(REF).read().split("\n")
    i0 = next(i for i, l in enumerate(src) if l.strip() == "# Compute TBN matrix")
    i1 = next(i for i in range(i0, len(src)) if src[i].strip() == "primrot = (")
    i2 = next(i for i in range(i1, len(src)) if src[i].strip() == ")")
    body = [l[8:] for l in src[i0 + 1:i2 + 1]]
    assert body[0].startswith("tangent = vcenterdu") and len(body) == 13, body
    return "\n".join(body)


if __name__ == "__main__":
    assert os.path.exists(REF), "the reference is only mounted in the build container"
    code = reference_statements()
    g = torch.Generator().manual_seed(23)
    out = {}
    for tag, B, n in (("a", 2, 7), ("b", 1, 4)):
        du = torch.randn(B, n, n, 3, generator=g).float().double()
        dv = torch.randn(B, n, n, 3, generator=g).float().double()
        if tag == "b":
            du[0, 0, 0] = 0.0                       # |du| = 0: clamp
            dv[0, 1, 1] = 2.0 * du[0, 1, 1]         # dv parallel to du: |t x dv| ~ 0: clamp
        du.requires_grad_(True), dv.requires_grad_(True)
        env = {"torch": torch, "vcenterdu": du, "vcenterdv": dv, "expr_encoding": types.SimpleNamespace(size=lambda i: B)}
        exec(code, env)
        primrot = env["primrot"]
        w = torch.randn(primrot.shape, generator=g).float().double()
        (w * primrot).sum().backward()
        for name, t in (("du", du), ("dv", dv), ("primrot", primrot), ("g_primrot", w), ("g_du", du.grad), ("g_dv", dv.grad)):
            out["%s_%s" % (tag, name)] = t.detach().numpy()
    np.savez_compressed(os.path.join(HERE, "primframe.npz"), **out)
    print("wrote primframe.npz:", {k: v.shape for k, v in out.items() if k.startswith("a_")})
