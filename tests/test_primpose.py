"""Residual composition of the decoder -> raymarch hand-off (row N2; models/decoders/assembler.py:241-253 + models/utils.py
Rodrigues).  The oracle (oracle/primpose_oracle.py, numpy f64, hand-written gradients) is pinned to vectors made with the
reference's own Rodrigues module and autograd (tests/golden/primpose.npz, tests/golden/gen_primpose.py); the gfx950 kernels
(csrc/primpose.hip, fp32) are compared with the oracle and with the golden vectors: 2e-6 of the largest magnitude forward,
2e-5 backward (fp32 sin / cos / 1 / theta chains; the golden inputs are float32 values held in f64)."""
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ("a", "b", "c", "d")
INS = ("pos0", "rot0", "scale0", "posres", "rotres", "scaleres")
GRADS = ("g_pos0", "g_rot0", "g_posres", "g_rotres", "g_scaleres")


def _case(tag):
    g = np.load(os.path.join(GOLDEN, "primpose.npz"))
    return {k[len(tag) + 1:]: g[k] for k in g.files if k.startswith(tag + "_")}


@pytest.mark.parametrize("tag", CASES)
def test_oracle_matches_the_reference_vectors(tag):
    from oracle import primpose_oracle as po
    c = _case(tag)
    ins = [c[k] for k in INS]
    pos, rot, scale = po.prim_residuals(*ins, float(c["rw"]))
    assert np.abs(pos - c["primpos"]).max() <= 1e-14 and np.abs(rot - c["primrot"]).max() <= 1e-14
    assert np.abs(np.broadcast_to(scale, c["primscale"].shape) - c["primscale"]).max() <= 1e-14
    grads = po.prim_residuals_backward(*ins, float(c["rw"]), c["g_primpos"], c["g_primrot"], c["g_primscale"])
    for name, got in zip(GRADS, grads):
        assert got.shape == c[name].shape, name
        assert np.abs(got - c[name]).max() <= 1e-12 * max(1.0, np.abs(c[name]).max()), name


def test_oracle_rodrigues_is_a_rotation_and_handles_zero():
    from oracle import primpose_oracle as po
    v = np.concatenate([np.random.default_rng(3).normal(size=(50, 3)), np.zeros((2, 3))])
    R = po.rodrigues(v)
    # theta = sqrt(1e-5 + |v|^2) > |v|: |a| < 1, so R is a rotation only up to ~1e-5 / |v|^2 -- the reference's own property
    big = (v ** 2).sum(-1) >= 0.25
    err = np.abs(np.matmul(R, np.swapaxes(R, -1, -2)) - np.eye(3)).max(axis=(-1, -2))
    assert big.sum() >= 20 and (err[big] <= 1e-3).all() and np.isfinite(R).all()
    assert np.abs(R[-1] - np.eye(3) * np.cos(np.sqrt(1e-5))).max() <= 1e-15   # zero vector: cos(theta) I


@pytest.mark.gpu
@pytest.mark.parametrize("tag", CASES)
def test_kernels_match_the_reference_vectors(tag):
    from ava256_amd.placement import prim_residuals
    c = _case(tag)
    dev = "cuda"
    t = {k: torch.from_numpy(c[k]).float().to(dev) for k in INS}
    for k in INS:
        if k != "scale0":
            t[k].requires_grad_(True)
    N = c["primpos"].shape[0]
    pos, rot, scale = prim_residuals(t["pos0"], t["rot0"], t["scale0"], t["posres"], t["rotres"], t["scaleres"], float(c["rw"]), N)
    for got, name in ((pos, "primpos"), (rot, "primrot"), (scale, "primscale")):
        ref = c[name]
        assert tuple(got.shape) == ref.shape
        assert np.abs(got.detach().cpu().numpy() - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max()), name
    gp, gr, gs = (torch.from_numpy(c[k]).float().to(dev) for k in ("g_primpos", "g_primrot", "g_primscale"))
    ((gp * pos).sum() + (gr * rot).sum() + (gs * scale).sum()).backward()
    for name, k in zip(GRADS, ("pos0", "rot0", "posres", "rotres", "scaleres")):
        got, ref = t[k].grad.cpu().numpy(), c[name]
        assert got.shape == ref.shape, name
        assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), (name, np.abs(got - ref).max())


@pytest.mark.gpu
def test_kernels_match_the_oracle_at_size_and_partial_outputs():
    """C3-sized K with every input shared by the frames except the position (the stand-in decoder's case), ragged K; a loss
    that uses only primrot (the other output gradients arrive as None)."""
    from oracle import primpose_oracle as po
    from ava256_amd.placement import prim_residuals
    rng = np.random.default_rng(8)
    N, K = 4, 16384 + 37
    pos0 = rng.normal(size=(N, K, 3)).astype(np.float32)
    rot0 = np.linalg.qr(rng.normal(size=(K, 3, 3)))[0].astype(np.float32)
    scale0 = (np.abs(rng.normal(size=(K, 1))) + 0.5).astype(np.float32)
    posres, rotres = (0.01 * rng.normal(size=(K, 3))).astype(np.float32), (0.3 * rng.normal(size=(K, 3))).astype(np.float32)
    scaleres = (1 + 0.1 * rng.normal(size=(K, 3))).astype(np.float32)
    ins = [pos0, rot0, scale0, posres, rotres, scaleres]
    dev = "cuda"
    t = [torch.from_numpy(a).to(dev) for a in ins]
    for i in (0, 3, 4, 5):
        t[i].requires_grad_(True)
    pos, rot, scale = prim_residuals(*t, 0.6, N)
    rp, rr, rs = po.prim_residuals(*[a.astype(np.float64) for a in ins], 0.6)
    assert np.abs(pos.detach().cpu().numpy() - rp).max() <= 2e-6 * np.abs(rp).max()
    assert np.abs(rot.detach().cpu().numpy() - np.broadcast_to(rr, (N, K, 3, 3))).max() <= 2e-6
    assert np.abs(scale.detach().cpu().numpy() - np.broadcast_to(rs, (N, K, 3))).max() <= 2e-6 * np.abs(rs).max()
    gr = rng.normal(size=(N, K, 3, 3)).astype(np.float32)
    (torch.from_numpy(gr).to(dev) * rot).sum().backward()
    z3 = np.zeros((N, K, 3))
    ref = po.prim_residuals_backward(*[a.astype(np.float64) for a in ins], 0.6, z3, gr.astype(np.float64), z3)
    assert not t[0].grad.any() and not t[3].grad.any() and not t[5].grad.any()
    got = t[4].grad.cpu().numpy()
    assert got.shape == ref[3].shape and np.abs(got - ref[3]).max() <= 2e-5 * np.abs(ref[3]).max()
    assert t[1].grad is None   # rot0 did not ask for a gradient


@pytest.mark.gpu
def test_bad_arguments_are_refused():
    from ava256_amd.placement import prim_residuals
    dev = "cuda"
    K, N = 8, 2
    ok = dict(pos0=torch.zeros(N, K, 3, device=dev), rot0=torch.eye(3, device=dev).expand(K, 3, 3).contiguous(),
              scale0=torch.ones(K, 1, device=dev), posres=torch.zeros(K, 3, device=dev), rotres=torch.zeros(K, 3, device=dev),
              scaleres=torch.ones(K, 3, device=dev))
    pos, rot, scale = prim_residuals(*ok.values(), 1.0, N)
    assert torch.equal(pos, ok["pos0"]) and scale.shape == (N, K, 3) and float(scale.min()) == 1.0
    with pytest.raises(RuntimeError):
        prim_residuals(*{**ok, "rotres": torch.zeros(3, K, 3, device=dev)}.values(), 1.0, N)      # wrong frame count
    with pytest.raises(RuntimeError):
        prim_residuals(*{**ok, "posres": torch.zeros(K, 3, device=dev, dtype=torch.float64)}.values(), 1.0, N)
    with pytest.raises(RuntimeError):
        prim_residuals(*{**ok, "pos0": torch.zeros(N, K, 3)}.values(), 1.0, N)                     # CPU tensor: no CPU path
