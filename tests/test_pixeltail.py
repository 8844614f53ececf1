"""The fused decode tail (csrc/pixeltail.hip): colour calibration + matting + L1 image loss, one pass each way.

tests/golden/pixeltail.npz was made with the reference's own `Colorcal` module and `mean_ell_1` (tests/golden/gen_pixeltail.py).
CPU: the numpy checker reproduces it.  GPU: the kernels reproduce it -- irgbrec bit for bit -- and agree with the eager
statements on ragged shapes, with every optional input absent in turn."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle.pixeltail_oracle import decode_tail as o_fwd, decode_tail_backward as o_bwd

G = np.load(os.path.join(GOLDEN, "pixeltail.npz"))


def test_oracle_matches_the_reference_modules():
    rgba = G["rayrgba"].astype(np.float32)
    rgb, alpha = np.ascontiguousarray(rgba[..., :3].transpose(0, 3, 1, 2)), np.ascontiguousarray(rgba[..., 3:].transpose(0, 3, 1, 2))
    out, l1 = o_fwd(rgb, alpha, G["w"], G["b"], G["bg"], G["target"])
    np.testing.assert_array_equal(out.astype(np.float32), G["irgbrec"])          # float32 numpy = float32 torch, op for op
    assert abs(l1 - G["l1"]) <= 1e-6 * G["l1"]
    g_rgb, g_alpha, g_w, g_b, g_bg = o_bwd(rgb.astype(np.float64), alpha.astype(np.float64), G["w"].astype(np.float64),
                                            G["bg"].astype(np.float64), G["target"].astype(np.float64),
                                            G["irgbrec"].astype(np.float64), G["gup"].astype(np.float64), float(G["l1_weight"]))
    got = np.concatenate([g_rgb, g_alpha], 1).transpose(0, 2, 3, 1)
    for mine, ref in ((got, G["grad_rayrgba"]), (g_bg, G["grad_bg"]), (g_w, G["grad_w"]), (g_b, G["grad_b"])):
        assert np.abs(mine - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())


def _eager(rgba, w, b, bg, target):
    rgb = rgba.permute(0, 3, 1, 2)[:, :3].contiguous()
    alpha = rgba.permute(0, 3, 1, 2)[:, 3:4].contiguous()
    out = rgb
    if w is not None:
        out = w[:, :, None, None] * out + b[:, :, None, None]
    if bg is not None:
        out = out + (1.0 - alpha) * bg
    l1 = None if target is None else (out - target).abs().sum()
    return out, alpha, l1


@pytest.mark.gpu
def test_kernels_match_the_reference_fixture():
    from ava256_amd.pixeltail import decode_tail
    t = {k: torch.from_numpy(G[k]).cuda() for k in ("rayrgba", "w", "b", "bg", "target", "gup")}
    for k in ("rayrgba", "w", "b", "bg"):
        t[k].requires_grad_(True)
    irgbrec, ialpha, l1sum = decode_tail(t["rayrgba"], t["w"], t["b"], t["bg"], t["target"])
    assert torch.equal(irgbrec.cpu(), torch.from_numpy(G["irgbrec"]))            # bit-identical to the eager statements
    assert torch.equal(ialpha[:, 0].cpu(), torch.from_numpy(G["rayrgba"][..., 3]))
    l1 = l1sum / irgbrec.numel()
    assert abs(float(l1) - float(G["l1"])) <= 1e-6 * float(G["l1"])
    (float(G["l1_weight"]) * l1 + (t["gup"] * irgbrec).sum()).backward()
    for k, ref in (("rayrgba", G["grad_rayrgba"]), ("bg", G["grad_bg"]), ("w", G["grad_w"]), ("b", G["grad_b"])):
        got = t[k].grad.cpu().numpy()
        assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), k


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 37, 53), (1, 128, 128), (5, 16, 300)])
@pytest.mark.parametrize("have", ["all", "no_bg", "no_cal", "no_target", "bare"])
def test_kernels_match_the_eager_statements(shape, have):
    """Ragged sizes (not multiples of the 256-pixel workgroup), each optional input absent in turn, an upstream gradient on
    irgbrec AND on ialpha next to the loss: forward bit-identical, gradients to fp32 summation accuracy."""
    from ava256_amd.pixeltail import decode_tail
    N, H, W = shape
    g = torch.Generator(device="cuda").manual_seed(N * 100 + W)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    rgba = torch.cat([100 + 40 * r(N, H, W, 3), torch.rand(N, H, W, 1, device="cuda", generator=g)], -1).contiguous()
    w = (1 + 0.1 * r(N, 3)) if have not in ("no_cal", "bare") else None
    b = (2.0 * r(N, 3)) if w is not None else None
    bg = (100 + 30 * r(N, 3, H, W)) if have not in ("no_bg", "bare") else None
    target = (100 + 40 * r(N, 3, H, W)) if have not in ("no_target", "bare") else None
    gup, gal = 0.01 * r(N, 3, H, W), 0.02 * r(N, 1, H, W)
    leaves = {}
    for nm, v in (("rgba", rgba), ("w", w), ("b", b), ("bg", bg)):
        leaves[nm] = None if v is None else v.clone().requires_grad_(True)
    ref_out, ref_alpha, ref_l1 = _eager(leaves["rgba"], leaves["w"], leaves["b"], leaves["bg"], target)
    loss = (gup * ref_out).sum() + (gal * ref_alpha).sum() + (0.7 * ref_l1 if ref_l1 is not None else 0.0)
    loss.backward()
    ref_g = {k: (None if v is None else v.grad.clone()) for k, v in leaves.items()}
    mine = {k: (None if v is None else v.detach().clone().requires_grad_(True)) for k, v in leaves.items()}
    out, alpha, l1sum = decode_tail(mine["rgba"], mine["w"], mine["b"], mine["bg"], target)
    assert torch.equal(out, ref_out.detach()) and torch.equal(alpha, ref_alpha.detach())
    if target is not None:
        assert abs(float(l1sum) - float(ref_l1)) <= 2e-6 * float(ref_l1)
    ((gup * out).sum() + (gal * alpha).sum() + 0.7 * l1sum).backward()
    for k in mine:
        if mine[k] is not None:
            a, bb = mine[k].grad, ref_g[k]
            assert (a - bb).abs().max().item() <= 2e-5 * max(1.0, bb.abs().max().item()), k


@pytest.mark.gpu
def test_decode_tail_argument_errors():
    from ava256_amd.pixeltail import decode_tail
    x = torch.zeros(2, 8, 8, 4, device="cuda")
    with pytest.raises(RuntimeError):
        decode_tail(x.cpu())
    with pytest.raises(RuntimeError):
        decode_tail(x, cw=torch.ones(2, 3, device="cuda"))                 # cw without cb
    with pytest.raises(RuntimeError):
        decode_tail(x, bg=torch.zeros(2, 3, 8, 9, device="cuda"))
    with pytest.raises(RuntimeError):
        decode_tail(torch.zeros(2, 8, 8, 3, device="cuda"))
