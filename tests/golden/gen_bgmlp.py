"""Golden vectors for the background MLP: the reference's own BackgroundModelSimple
(/root/reference/models/bg/mlp2d.py:14-72) run here on CPU in float32 with seeded weights.
Output: tests/golden/bgmlp.npz = inputs, every parameter, the output and the gradients of a fixed linear loss.
Run in the build container only (needs /root/reference):  python tests/golden/gen_bgmlp.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
from models.bg.mlp2d import BackgroundModelSimple  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    torch.manual_seed(77)
    ncams, nident, B, H, W = 3, 2, 2, 12, 20
    m = BackgroundModelSimple(ncams, nident)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():  # initseq leaves small weights and zero biases: make every term of the forward matter
        for p in m.parameters():
            if p.dim() == 1:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
    camindex = torch.tensor([2, 0])
    idindex = torch.tensor([1, 1])
    samplecoords = torch.rand(B, H, W, 2, generator=g) * 2 - 1
    gout = torch.randn(B, 3, H, W, generator=g)
    bg = m(camindex, idindex, samplecoords)
    (bg * gout).sum().backward()
    out = dict(camindex=camindex.numpy(), idindex=idindex.numpy(), samplecoords=samplecoords.numpy(), gout=gout.numpy(),
               bg=bg.detach().numpy())
    for k, v in m.state_dict().items():
        out["param/" + k] = v.numpy()
    for k, p in m.named_parameters():
        out["grad/" + k] = p.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "bgmlp.npz"), **out)
    print("bgmlp.npz:", {k: v.shape for k, v in out.items() if not k.startswith("param/mlp")})

    # Second fixture, SAME module (parameters are those of bgmlp.npz and are not stored again): two 64 x 64 images =
    # 2 x 16 tiles of the fused kernels' 256 pixels, the second image with sample coordinates in [-2.5, 2.5] (cropped /
    # jittered pixelcoords under autoencoder.py:231-237).  Gradients of the large square layers are kept for two of the
    # four (1 MB less), every other gradient in full.
    m.zero_grad()
    g2 = torch.Generator().manual_seed(6)
    B, H, W = 2, 64, 64
    camindex, idindex = torch.tensor([1, 2]), torch.tensor([0, 1])
    samplecoords = torch.rand(B, H, W, 2, generator=g2) * 2 - 1
    samplecoords[1] *= 2.5
    gout = torch.randn(B, 3, H, W, generator=g2)
    bg = m(camindex, idindex, samplecoords)
    (bg * gout).sum().backward()
    out2 = dict(camindex=camindex.numpy(), idindex=idindex.numpy(), samplecoords=samplecoords.numpy(), gout=gout.numpy(),
                bg=bg.detach().numpy())
    for k, p in m.named_parameters():
        if k not in ("mlp.4.weight", "mlp.8.weight"):
            out2["grad/" + k] = p.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "bgmlp_multitile.npz"), **out2)
    print("bgmlp_multitile.npz:", {k: v.shape for k, v in out2.items() if not k.startswith("grad/mlp")})

    # Third fixture (round 6), SAME module again: ONE ragged image of 29 x 31 = 899 pixels = three full tiles of the fused
    # kernels and a fourth with 131 valid rows -- the non-zero hidden biases of the first fixture's parameters apply.
    m.zero_grad()
    g3 = torch.Generator().manual_seed(7)
    B, H, W = 1, 29, 31
    camindex, idindex = torch.tensor([0]), torch.tensor([1])
    samplecoords = torch.rand(B, H, W, 2, generator=g3) * 2 - 1
    gout = torch.randn(B, 3, H, W, generator=g3)
    bg = m(camindex, idindex, samplecoords)
    (bg * gout).sum().backward()
    out3 = dict(camindex=camindex.numpy(), idindex=idindex.numpy(), samplecoords=samplecoords.numpy(), gout=gout.numpy(),
                bg=bg.detach().numpy())
    for k, p in m.named_parameters():
        if k not in ("mlp.4.weight", "mlp.8.weight"):
            out3["grad/" + k] = p.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "bgmlp_ragged.npz"), **out3)
    print("bgmlp_ragged.npz:", {k: v.shape for k, v in out3.items() if not k.startswith("grad/mlp")})


if __name__ == "__main__":
    main()
