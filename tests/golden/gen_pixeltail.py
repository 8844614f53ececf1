"""tests/golden/gen_pixeltail.py -- golden vectors of the decode tail (colour calibration + matting + L1 image loss) made
with the reference's OWN modules: `Colorcal` (models/colorcals/colorcal.py) and `mean_ell_1` (losses.py) are imported from
the mounted reference, the matting statement is the one of models/autoencoder.py:264, autograd gives the gradients.
float32 (the arithmetic type of the path), seeded inputs.  Run in the build container; writes tests/golden/pixeltail.npz."""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

if __name__ == "__main__":
    assert os.path.isdir(REF), "the reference is only mounted in the build container"
    sys.path.insert(0, REF)
    from losses import mean_ell_1                       # losses.py:12-14
    from models.colorcals.colorcal import Colorcal      # colorcal.py:11-31
    g = torch.Generator().manual_seed(77)
    N, H, W, ncams, nident = 3, 21, 19, 5, 2
    cc = Colorcal(ncams, nident)
    with torch.no_grad():
        cc.wcam.add_(0.1 * torch.randn(ncams, 3, generator=g)), cc.bcam.add_(2.0 * torch.randn(ncams, 3, generator=g))
        cc.wident.add_(0.05 * torch.randn(nident, 3, generator=g)), cc.bident.add_(1.0 * torch.randn(nident, 3, generator=g))
    camindex, idindex = torch.tensor([4, 0, 2]), torch.tensor([1, 0, 1])
    rayrgba = torch.cat([100 + 40 * torch.randn(N, H, W, 3, generator=g), torch.rand(N, H, W, 1, generator=g)], -1)
    rayrgba[0, :3, :3, 3] = 1.0                                        # saturated rays: (1 - alpha) = 0 exactly
    bg = (100 + 30 * torch.randn(N, 3, H, W, generator=g)).requires_grad_(True)
    target = 100 + 40 * torch.randn(N, 3, H, W, generator=g)
    rgba = rayrgba.clone().requires_grad_(True)
    rayrgb = rgba.permute(0, 3, 1, 2)[:, :3].contiguous()              # mvpraymarcher.py:50
    rayalpha = rgba.permute(0, 3, 1, 2)[:, 3:4].contiguous()           # mvpraymarcher.py:51
    out = cc(rayrgb, camindex, idindex)                                # autoencoder.py:254-256
    irgbrec = out + (1.0 - rayalpha) * bg                              # autoencoder.py:264
    target[1, :, 5, 5] = irgbrec.detach()[1, :, 5, 5]                  # |0|: the sign(0) = 0 convention of torch.abs
    loss = mean_ell_1(irgbrec, target)                                 # ddp-train.py:404-405
    gup = 0.01 * torch.randn(N, 3, H, W, generator=g)                  # an extra upstream gradient on irgbrec
    (3.0 * loss + (gup * irgbrec).sum()).backward()
    w = (cc.wcam[camindex] + cc.wident[idindex]).detach()
    b = (cc.bcam[camindex] + cc.bident[idindex]).detach()
    # gradients of the per-image (w, b) -- what the parameters' index-adds scatter -- from a second autograd pass
    w_leaf, b_leaf = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    rgba2 = rayrgba.clone()
    out2 = w_leaf[:, :, None, None] * rgba2.permute(0, 3, 1, 2)[:, :3] + b_leaf[:, :, None, None]
    ir2 = out2 + (1.0 - rgba2.permute(0, 3, 1, 2)[:, 3:4]) * bg.detach()
    (3.0 * mean_ell_1(ir2, target) + (gup * ir2).sum()).backward()
    assert torch.equal(ir2.detach(), irgbrec.detach())
    np.savez_compressed(os.path.join(OUT, "pixeltail.npz"), rayrgba=rayrgba.numpy(), w=w.numpy(), b=b.numpy(),
                        bg=bg.detach().numpy(), target=target.numpy(), gup=gup.numpy(), l1_weight=np.float32(3.0),
                        irgbrec=irgbrec.detach().numpy(), l1=loss.detach().numpy(), grad_rayrgba=rgba.grad.numpy(),
                        grad_bg=bg.grad.numpy(), grad_w=w_leaf.grad.numpy(), grad_b=b_leaf.grad.numpy())
    print("pixeltail.npz written: l1 = %.6f" % float(loss.detach()))
