"""Shared helpers of the GPU parity tests (test infrastructure)."""
import json
import os

import numpy as np
import torch

from conftest import GOLDEN


def to_dev(x, dev="cuda"):
    return torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32).to(dev).contiguous()


def npf(t):
    return t.detach().cpu().numpy().astype(np.float64)


def cosine(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    den = np.sqrt((a * a).sum() * (b * b).sum())
    return 1.0 if den == 0 else float((a * b).sum() / den)


def load_krt_400940():
    """Camera 400940 exactly as the reference's fixture derives it (tests/test_extensions.py:25-41,
    utils.py:142-170): intrin = K^T, extrin = T[:4,:3]^T, campos = -R^T t."""
    item = json.load(open(os.path.join(GOLDEN, "camera_400940.json")))["KRT"][0]
    intrin = np.array(item["K"]).T
    extrin = np.array(item["T"])[:4, :3].T
    campos = (-np.dot(extrin[:3, :3].T, extrin[:3, 3])).astype(np.float32)[None]
    camrot = extrin[:3, :3].astype(np.float32)[None]
    focal = np.diag(intrin[:2, :2]).astype(np.float32)[None]
    princpt = intrin[:2, 2].astype(np.float32)[None]
    return campos, camrot, focal, princpt


def scene_rays(oracle, s):
    """Rays of a synthetic scene computed by the oracle (float64)."""
    return oracle.raydirs(s["campos"].numpy(), s["camrot"].numpy(), s["focal"].numpy(), s["princpt"].numpy(),
                          s["pixelcoords"].numpy(), s["volradius"])


def make_placement_inputs(nprims, B=None, V=7306, T=1024, seed=77):
    """Seeded inputs of the primitive-placement tests (shared by tests/golden/gen_placement.py, which feeds them to the
    reference's own statements): geo [B,V,3] in millimetre-like units, idxim [T,T,3] vertex indices, barim [T,T,3]
    barycentric weights, volradius, and three weight arrays for a scalar loss."""
    rng = np.random.default_rng(seed + nprims)
    B = B or (2 if nprims == 16384 else 3)
    geo = (rng.normal(size=(B, V, 3)) * 60.0).astype(np.float32)
    idxim = rng.integers(0, V, size=(T, T, 3), dtype=np.int64)
    bar = rng.random(size=(T, T, 3)).astype(np.float32) + np.float32(0.05)
    barim = (bar / bar.sum(-1, keepdims=True)).astype(np.float32)
    ny, nx = (16, 16) if nprims == 256 else (128, 128)
    w = (rng.normal(size=(B, nprims, 3)).astype(np.float32), rng.normal(size=(B, ny, nx, 3)).astype(np.float32),
         rng.normal(size=(B, ny, nx, 3)).astype(np.float32))
    return geo, idxim, barim, 256.0, w
