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
