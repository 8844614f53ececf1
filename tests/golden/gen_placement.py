#!/usr/bin/env python3
"""Generate tests/golden/placement_ref.npz from the REFERENCE'S OWN LINES.

Runs only in the build container.  The text of /root/reference/models/decoders/assembler.py is read at run time and the
statements that build `postex` (assembler.py:118-122) and that read it for 256 and 16384 primitives (primpos, geodu,
geodv, vcenterdu, vcenterdv: assembler.py:143-144,166-170 and :180-181,202-206) are executed as they stand, in float32
on CPU (the reference's dtype), with a stand-in `self` carrying seeded `idxim` / `barim` / `volradius`.  Autograd then
gives d/d geo of a seeded weighted sum of the three outputs.  Nothing of the reference is stored; the inputs are
regenerated from seeds by `make_inputs` (shared with the tests), the fixture holds only the outputs.
"""
import os
import re
import sys
import types

import numpy as np
import torch

REF = "/root/reference/models/decoders/assembler.py"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from helpers import make_placement_inputs  # noqa: E402


def reference_statements():
    src = open(REF).read().split("\n")
    # the postex expression: from "postex = (" to its closing ") .permute(...) / self.volradius"
    i0 = next(i for i, l in enumerate(src) if l.strip() == "postex = (")
    i1 = next(i for i in range(i0, len(src)) if ".permute(0, 3, 1, 2) / self.volradius" in src[i])
    postex = "\n".join(l[8:] for l in src[i0:i1 + 1])
    branches = {}
    for n in (256, 16384):
        j0 = next(i for i, l in enumerate(src) if l.strip() == "elif self.nprims == %d:" % n)
        j1 = next(i for i in range(j0 + 1, len(src)) if src[i].strip().startswith("elif self.nprims =="))
        body = src[j0 + 1:j1]
        keep = [l for l in body if re.match(r"\s+(primpos|geodu|geodv|vcenterdu|vcenterdv) = ", l)]
        assert len(keep) == 5, keep
        branches[n] = "\n".join(l[12:] for l in keep)
    return postex, branches


def main():
    postex_src, branches = reference_statements()
    out = {}
    for nprims in (256, 16384):
        geo_np, idxim_np, barim_np, volradius, w = make_placement_inputs(nprims)
        self = types.SimpleNamespace(idxim=torch.from_numpy(idxim_np).long(), barim=torch.from_numpy(barim_np),
                                     volradius=volradius, nprims=nprims)
        geo = torch.from_numpy(geo_np.copy()).requires_grad_(True)
        ns = {"self": self, "geo": geo, "torch": torch, "nprims": nprims}
        exec(postex_src, ns)
        exec(branches[nprims], ns)
        primpos, du, dv = ns["primpos"], ns["vcenterdu"], ns["vcenterdv"]
        loss = (primpos * torch.from_numpy(w[0])).sum() + (du * torch.from_numpy(w[1])).sum() + \
               (dv * torch.from_numpy(w[2])).sum()
        loss.backward()
        tag = "k%d_" % nprims
        out[tag + "primpos"] = primpos.detach().numpy()
        out[tag + "vcenterdu"] = du.detach().numpy()
        out[tag + "vcenterdv"] = dv.detach().numpy()
        out[tag + "grad_geo"] = geo.grad.numpy()
        print(nprims, primpos.shape, du.shape, "postex", tuple(ns["postex"].shape), "|grad|", float(geo.grad.abs().max()))
    np.savez_compressed(os.path.join(HERE, "placement_ref.npz"), **out)


if __name__ == "__main__":
    main()
