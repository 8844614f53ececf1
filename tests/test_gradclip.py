"""Row N4: multi-tensor NaN/Inf masking + global-norm clipping (ddp-train.py:434-441).

CPU: the oracle is pinned against the reference's own statements executed with torch on CPU tensors.
GPU: the HIP passes (through the C ABI) against the oracle."""
import numpy as np
import pytest
import torch

from oracle.gradclip_oracle import sanitize_and_clip

SIZES = [1, 3, 4, 5, 63, 64, 65, 1000, 16384, 16385, 40001, 65536, 65537, 3 * 65536 + 7, 200000]


def make_grads(seed, sizes=SIZES, scale=1.0, bad=True):
    rng = np.random.default_rng(seed)
    gs = []
    for i, n in enumerate(sizes):
        g = (rng.normal(size=n) * scale * (1 + i % 3)).astype(np.float32)
        if bad and n > 2:
            idx = rng.choice(n, size=max(1, n // 37), replace=False)
            kinds = rng.integers(0, 3, size=idx.size)
            g[idx[kinds == 0]] = np.nan
            g[idx[kinds == 1]] = np.inf
            g[idx[kinds == 2]] = -np.inf
        gs.append(g)
    return gs


def reference_statements(grads, max_norm):
    """ddp-train.py:436-441 verbatim in meaning, on CPU tensors."""
    params = [torch.nn.Parameter(torch.zeros(g.shape)) for g in grads]
    for p, g in zip(params, grads):
        p.grad = torch.from_numpy(g.copy())
    for p in params:
        p.grad.data[torch.isnan(p.grad.data)] = 0
        p.grad.data[torch.isinf(p.grad.data)] = 0
    total = torch.nn.utils.clip_grad_norm_(params, max_norm)
    return [p.grad.numpy() for p in params], float(total)


@pytest.mark.parametrize("max_norm", [1.0, 1.0e9, 0.0])
def test_oracle_matches_reference_statements(max_norm):
    grads = make_grads(11)
    want, want_norm = reference_statements(grads, max_norm)
    got, norm = sanitize_and_clip(grads, max_norm)
    assert abs(norm - want_norm) <= 2e-6 * max(1.0, want_norm)
    for a, b in zip(got, want):
        assert a.shape == b.shape and np.all(np.isfinite(a))
        np.testing.assert_array_equal(a == 0, b == 0)       # exactly the same elements were zeroed
        np.testing.assert_allclose(a, b, rtol=2e-6, atol=0)  # fp32 norm rounding only


def test_oracle_empty_and_clean():
    got, norm = sanitize_and_clip([], 1.0)
    assert got == [] and norm == 0.0
    g = make_grads(3, sizes=[10, 20], bad=False)
    got, norm = sanitize_and_clip(g, 1e9)
    for a, b in zip(got, g):
        np.testing.assert_array_equal(a, b)  # no clipping, nothing to sanitise: untouched


@pytest.mark.gpu
@pytest.mark.parametrize("max_norm", [1.0, 1.0e9, 0.0])
@pytest.mark.parametrize("bad", [True, False])
def test_hip_matches_oracle(max_norm, bad):
    from ava256_amd.gradclip import GradClipper
    grads = make_grads(5, bad=bad)
    want, want_norm = sanitize_and_clip(grads, max_norm)
    dev = torch.device("cuda", 0)
    t = [torch.from_numpy(g.copy()).to(dev) for g in grads]
    clipper = GradClipper(dev)
    norm = clipper(t, max_norm)
    torch.cuda.synchronize()
    assert abs(float(norm) - want_norm) <= 2e-6 * max(1.0, want_norm)
    for a, b in zip(t, want):
        a = a.cpu().numpy()
        np.testing.assert_array_equal(a == 0, b == 0)
        np.testing.assert_allclose(a, b, rtol=2e-6, atol=0)
    if not bad and max_norm == 1.0e9:  # nothing to do: bit-identical to the input
        for a, g in zip(t, grads):
            np.testing.assert_array_equal(a.cpu().numpy(), g)


@pytest.mark.gpu
def test_hip_many_tensors_views_and_parameters():
    """More tensors than one launch carries (48), an unaligned view, Parameters with and without .grad."""
    from ava256_amd.gradclip import GradClipper, sanitize_and_clip_
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(9)
    sizes = [int(s) for s in rng.integers(1, 5000, size=131)]
    grads = make_grads(21, sizes=sizes)
    want, want_norm = sanitize_and_clip(grads, 2.5)
    params = []
    for g in grads:
        p = torch.nn.Parameter(torch.zeros(g.shape, device=dev))
        p.grad = torch.from_numpy(g.copy()).to(dev)
        params.append(p)
    params.insert(7, torch.nn.Parameter(torch.zeros(5, device=dev)))  # no grad: skipped like the reference does
    norm = GradClipper(dev)(params, 2.5)
    torch.cuda.synchronize()
    assert abs(float(norm) - want_norm) <= 2e-6 * max(1.0, want_norm)
    got = [p.grad.cpu().numpy() for p in params if p.grad is not None]
    for a, b in zip(got, want):
        np.testing.assert_allclose(a, b, rtol=2e-6, atol=0)
    # a view starting 4 bytes into an allocation (not 16-byte aligned)
    base = torch.from_numpy(make_grads(2, sizes=[70001])[0]).to(dev)
    view = base[1:]
    want2, n2 = sanitize_and_clip([base.cpu().numpy()[1:]], 0.5)
    first = float(base[0])
    norm2 = sanitize_and_clip_([view], 0.5)
    torch.cuda.synchronize()
    np.testing.assert_allclose(view.cpu().numpy(), want2[0], rtol=2e-6, atol=0)
    assert abs(float(norm2) - n2) <= 2e-6 * max(1.0, n2)
    assert float(base[0]) == first or (np.isnan(first) and np.isnan(float(base[0])))  # element before the view untouched


@pytest.mark.gpu
def test_hip_rejects_wrong_inputs():
    from ava256_amd.gradclip import GradClipper
    dev = torch.device("cuda", 0)
    c = GradClipper(dev)
    with pytest.raises(RuntimeError):
        c([torch.zeros(4, dtype=torch.float64, device=dev)], 1.0)
    with pytest.raises(RuntimeError):
        c([torch.zeros(4, 4, device=dev).t()[1:]], 1.0)
    assert float(c([], 1.0)) == 0.0
    # the returned norm is a fresh scalar (like clip_grad_norm_'s): a later call must not change an earlier result
    g1, g2 = torch.full((10,), 3.0, device=dev), torch.full((10,), 0.5, device=dev)
    n1 = c([g1], 100.0)
    n2 = c([g2], 100.0)
    assert abs(float(n1) - 3.0 * 10 ** 0.5) < 1e-5 and abs(float(n2) - 0.5 * 10 ** 0.5) < 1e-6
