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


# ---- the TBN frame (assembler.py:227-240) --------------------------------------------------------------------------------
def _frame_case(tag):
    g = np.load(os.path.join(GOLDEN, "primframe.npz"))
    return {k[len(tag) + 1:]: g[k] for k in g.files if k.startswith(tag + "_")}


@pytest.mark.parametrize("tag", ("a", "b"))
def test_frame_oracle_matches_the_reference_lines(tag):
    """oracle/primpose_oracle.prim_frame against vectors made by executing assembler.py:227-240 as they stand
    (tests/golden/gen_primframe.py); case b holds a zero tangent difference and a dv parallel to du (the 1e-8 clamps)."""
    from oracle import primpose_oracle as po
    c = _frame_case(tag)
    B = c["du"].shape[0]
    rot = po.prim_frame(c["du"], c["dv"]).reshape(B, -1, 3, 3)
    # dv parallel to du (case b, texel (1, 1)): t x dv is rounding noise divided by the 1e-8 clamp -- whatever the cross
    # product's rounding leaves, different in every implementation, the reference's included: finite, not compared
    keep = np.ones(rot.shape[:2], dtype=bool)
    if tag == "b":
        keep[0, 5] = False
    assert np.abs(rot - c["primrot"])[keep].max() <= 1e-14 and np.isfinite(rot).all()
    gdu, gdv = po.prim_frame_backward(c["du"], c["dv"], c["g_primrot"].reshape(c["du"].shape[:-1] + (3, 3)))
    for got, name in ((gdu, "g_du"), (gdv, "g_dv")):
        ref, got = c[name].reshape(B, -1, 3), got.reshape(B, -1, 3)
        assert np.abs(got - ref)[keep].max() <= 1e-9 * max(1.0, np.abs(ref[keep]).max()), (name, np.abs(got - ref)[keep].max())
        assert np.isfinite(got).all()


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ("a", "b"))
def test_frame_kernels_match_the_reference_lines(tag):
    from ava256_amd.placement import prim_frame
    c = _frame_case(tag)
    du = torch.from_numpy(c["du"]).float().cuda().requires_grad_(True)
    dv = torch.from_numpy(c["dv"]).float().cuda().requires_grad_(True)
    rot = prim_frame(du, dv)
    assert tuple(rot.shape) == c["primrot"].shape
    # the clamped rows of case b divide by 1e-8: their entries are O(1e8 * eps) apart in fp32 -- compared relative to the row
    ref = c["primrot"]
    err = np.abs(rot.detach().cpu().numpy() - ref)
    clamped = np.zeros(ref.shape[:2], dtype=bool)
    if tag == "b":
        clamped[0, 0] = True      # |du| = 0
        clamped[0, 5] = True      # dv parallel to du (texel (1, 1) of a 4 x 4 grid)
    assert err[~clamped].max() <= 2e-6
    (torch.from_numpy(c["g_primrot"]).float().cuda() * rot).sum().backward()
    for t, name in ((du, "g_du"), (dv, "g_dv")):
        got, refg = t.grad.cpu().numpy().reshape(ref.shape[0], -1, 3), c[name].reshape(ref.shape[0], -1, 3)
        scale = np.abs(refg[~clamped]).max()
        assert np.abs(got - refg)[~clamped].max() <= 2e-5 * max(1.0, scale), name
        assert np.isfinite(got).all()


@pytest.mark.gpu
def test_placement_frame_residuals_chain_is_the_assembler():
    """prim_placement -> prim_frame -> prim_residuals = assembler.py:118-253 for 16384 primitives: against the oracles chained
    the same way, on a seeded mesh (the three kernels hand [B, K, .] tensors to each other without a copy)."""
    from oracle import placement_oracle as plo
    from oracle import primpose_oracle as po
    from ava256_amd.placement import prim_frame, prim_placement, prim_residuals
    from helpers import make_placement_inputs
    B, K = 2, 16384
    geo, idxim, barim, volradius, _ = make_placement_inputs(K, B=B, V=500, seed=4)
    dev = "cuda"
    g = torch.from_numpy(geo).to(dev).requires_grad_(True)
    pos0, du, dv = prim_placement(g, torch.from_numpy(idxim).to(dev), torch.from_numpy(barim).to(dev), volradius, K)
    rot0 = prim_frame(du, dv)
    rng = np.random.default_rng(2)
    posres, rotres = (0.01 * rng.normal(size=(B, K, 3))).astype(np.float32), (0.2 * rng.normal(size=(B, K, 3))).astype(np.float32)
    scaleres = (1 + 0.1 * rng.normal(size=(B, K, 3))).astype(np.float32)
    scale0 = torch.full((1,), 64.0, device=dev)                                   # assembler.py:173 `primscale = 64.0`
    pos, rot, scale = prim_residuals(pos0, rot0, scale0, *[torch.from_numpy(a).to(dev) for a in (posres, rotres, scaleres)], 0.5, B)
    rp, rdu, rdv = plo.placement(geo.astype(np.float64), idxim, barim.astype(np.float64), volradius, K)
    rrot0 = po.prim_frame(rdu, rdv).reshape(B, K, 3, 3)
    ep, er, es = po.prim_residuals(rp.reshape(B, K, 3), rrot0, 64.0, posres.astype(np.float64), rotres.astype(np.float64),
                                   scaleres.astype(np.float64), 0.5)
    assert np.abs(pos.detach().cpu().numpy() - ep).max() <= 1e-5 * np.abs(ep).max()
    assert np.abs(rot.detach().cpu().numpy() - er).max() <= 2e-4      # unit(du) of differences of nearby texels: fp32 cancellation
    assert np.abs(scale.detach().cpu().numpy() - es).max() <= 1e-5 * np.abs(es).max()
    (rot.sum() + pos.sum()).backward()
    assert torch.isfinite(g.grad).all() and float(g.grad.abs().max()) > 0
