"""Decoder -> raymarch hand-off (row N2): the numpy oracle is pinned to the layout the reference's real decoder modules
produce (tests/golden/assemble_map.npz), and the gfx950 kernel must equal the oracle BIT FOR BIT (pure data movement +
one multiply, one add, one max per element); its backward must equal autograd of the eager expression."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN


def test_oracle_layout_matches_the_reference_modules():
    from oracle import assemble_oracle as ao
    g = np.load(os.path.join(GOLDEN, "assemble_map.npz"))
    nh, B = int(g["nh"]), int(g["B"])
    d = g["rgb_dst"]                                     # flat index into [K,B,B,B,3]
    c = d % 3; x = (d // 3) % B; y = (d // (3 * B)) % B; z = (d // (3 * B * B)) % B; k = d // (3 * B ** 3)
    assert np.array_equal(ao.src_index_rgb(nh, B, k, z, y, x, c), g["rgb_src"])
    d = g["op_dst"]                                      # flat index into [K,B,B,B,1]
    x = d % B; y = (d // B) % B; z = (d // (B * B)) % B; k = d // (B ** 3)
    assert np.array_equal(ao.src_index_opacity(nh, B, k, z, y, x), g["op_src"])
    # the vectorised restatement agrees with the index functions (small config, every element)
    nh, B, N = 3, 4, 2
    S = nh * B
    rng = np.random.default_rng(1)
    tex = rng.normal(size=(N, 3 * B, S, S)).astype(np.float32)
    op = rng.normal(size=(N, B, S, S)).astype(np.float32)
    out = ao.assemble_template(tex, op, nh * nh, B)
    assert out.shape == (N, nh * nh, B, B, B, 4) and out.dtype == np.float32
    for n in range(N):
        for k in range(nh * nh):
            for z in range(B):
                for y in range(B):
                    for x in range(B):
                        for c in range(3):
                            v = tex[n].reshape(-1)[ao.src_index_rgb(nh, B, k, z, y, x, c)]
                            assert out[n, k, z, y, x, c] == max(np.float32(v * np.float32(25)) + np.float32(100), 0)
                        v = op[n].reshape(-1)[ao.src_index_opacity(nh, B, k, z, y, x)]
                        assert out[n, k, z, y, x, 3] == max(v, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(2, 3, 4), (1, 16, 8), (1, 128, 8)], ids=lambda c: "N%d_nh%d_B%d" % c)
def test_kernel_bit_exact_and_backward(cfg):
    from ava256_amd.assemble import assemble_template
    from oracle import assemble_oracle as ao
    N, nh, B = cfg
    S = nh * B
    g = torch.Generator(device="cuda").manual_seed(3)
    tex = (torch.randn(N, 3 * B, S, S, device="cuda", generator=g) * 2.0 - 3.5).requires_grad_(True)   # ~half below -4
    op = (torch.randn(N, B, S, S, device="cuda", generator=g)).requires_grad_(True)
    out = assemble_template(tex, op, nh * nh, B)
    ref = ao.assemble_template(tex.detach().cpu().numpy(), op.detach().cpu().numpy(), nh * nh, B)
    assert np.array_equal(out.detach().cpu().numpy(), ref)
    gout = torch.randn(out.shape, device="cuda", generator=g)
    out.backward(gout)
    # eager PyTorch statement of the same thing (rgb.py:137-143, geometry.py:183-185, assembler.py:261)
    t2, o2 = tex.detach().clone().requires_grad_(True), op.detach().clone().requires_grad_(True)
    rgb = t2.view(N, B, 3, nh, B, nh, B).permute(0, 3, 5, 1, 4, 6, 2).reshape(N, nh * nh, B, B, B, 3)
    a = o2.view(N, B, 1, nh, B, nh, B).permute(0, 3, 5, 1, 4, 6, 2).reshape(N, nh * nh, B, B, B, 1)
    eager = torch.cat([torch.relu(rgb * 25.0 + 100.0), torch.relu(a)], dim=-1)
    assert torch.equal(eager, out.detach())
    eager.backward(gout)
    assert torch.equal(t2.grad, tex.grad) and torch.equal(o2.grad, op.grad)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(3, 3, 4), (5, 16, 8), (4, 128, 8), (0, 3, 4)], ids=lambda c: "F%d_nh%d_B%d" % c)
def test_frame_broadcast_form_equals_assemble_times_gain(cfg):
    """mvp_template_assemble_frames_*: tplate[f] = gain[f] * assemble(tex, opacity) in one pass each way.  Forward bit-equal
    to the eager product (one rounding per element either way); backward against autograd of the eager statements
    (sums over frames / voxels in a different order: 1e-5 of the largest magnitude)."""
    from ava256_amd.assemble import assemble_template, assemble_template_frames
    F, nh, B = cfg
    S = nh * B
    g = torch.Generator(device="cuda").manual_seed(11)
    tex = (torch.randn(1, 3 * B, S, S, device="cuda", generator=g) * 2.0 - 3.5).requires_grad_(True)
    op = torch.randn(1, B, S, S, device="cuda", generator=g).requires_grad_(True)
    gain = (1.0 + 0.3 * torch.randn(F, device="cuda", generator=g)).requires_grad_(True)
    out = assemble_template_frames(tex, op, gain, nh * nh, B)
    assert out.shape == (F, nh * nh, B, B, B, 4)
    t2, o2, g2 = (t.detach().clone().requires_grad_(True) for t in (tex, op, gain))
    eager = g2.view(-1, 1, 1, 1, 1, 1) * assemble_template(t2, o2, nh * nh, B)
    assert torch.equal(out.detach(), eager.detach())
    gout = torch.randn(out.shape, device="cuda", generator=g)
    out.backward(gout)
    eager.backward(gout)
    # the oracle (numpy float64, oracle/assemble_oracle.py: pinned to the reference's decoders by assemble_map.npz and, for the
    # backward and the frame form, to autograd of the reference's statements by test_frame_oracle_equals_autograd...)
    from oracle import assemble_oracle as ao
    npf = lambda t: t.detach().cpu().numpy().astype(np.float64)   # noqa: E731
    ref = ao.assemble_template_frames(npf(tex), npf(op), npf(gain), nh * nh, B)
    assert ref.shape == tuple(out.shape)
    if F > 0:
        assert np.abs(npf(out) - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())
        rt, ro, rg = ao.assemble_template_frames_backward(npf(tex), npf(op), npf(gain), nh * nh, B, npf(gout))
        for got, want in ((tex.grad, rt), (op.grad, ro), (gain.grad, rg)):
            assert np.abs(npf(got) - want).max() <= 1e-5 * max(1.0, np.abs(want).max())
    for a, b in ((tex.grad, t2.grad), (op.grad, o2.grad), (gain.grad, g2.grad)):
        assert a.shape == b.shape and torch.isfinite(a).all()
        if F == 0:
            assert not a.any()
        else:
            assert (a - b).abs().max().item() <= 1e-5 * max(1.0, b.abs().max().item())


@pytest.mark.parametrize("cfg", [(3, 3, 4), (2, 4, 8)], ids=lambda c: "F%d_nh%d_B%d" % c)
def test_frame_oracle_equals_autograd_of_the_reference_statements(cfg):
    """oracle/assemble_oracle.assemble_template_backward / _frames / _frames_backward (hand-written) against torch autograd of
    the reference's statements (rgb.py:137-143, geometry.py:183-185, assembler.py:261 as trainloop.assemble_template_eager
    states them) times a per-frame gain, float64 on CPU."""
    from oracle import assemble_oracle as ao
    from ava256_amd.trainloop import assemble_template_eager
    F, nh, B = cfg
    S = nh * B
    rng = np.random.default_rng(5)
    tex, op = rng.normal(size=(1, 3 * B, S, S)) * 2 - 3.5, rng.normal(size=(1, B, S, S))
    gain, g = 1 + 0.3 * rng.normal(size=F), rng.normal(size=(F, nh * nh, B, B, B, 4))
    t, o, gn = (torch.tensor(a, requires_grad=True) for a in (tex, op, gain))
    out = gn.view(-1, 1, 1, 1, 1, 1) * assemble_template_eager(t, o, nh * nh, B)
    out.backward(torch.tensor(g))
    assert np.abs(ao.assemble_template_frames(tex, op, gain, nh * nh, B) - out.detach().numpy()).max() <= 1e-13
    gt, go, gg = ao.assemble_template_frames_backward(tex, op, gain, nh * nh, B, g)
    assert np.abs(gt - t.grad.numpy()).max() <= 1e-12 and np.abs(go - o.grad.numpy()).max() <= 1e-12
    assert np.abs(gg - gn.grad.numpy()).max() <= 1e-10 * max(1.0, np.abs(gg).max())
