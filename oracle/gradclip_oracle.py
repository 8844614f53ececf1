"""CPU checker for the multi-tensor gradient hygiene pass (row N4).  TEST INFRASTRUCTURE ONLY.

Restates /root/reference/ddp-train.py:434-441:
    for p in params:  p.grad.data[isnan] = 0 ; p.grad.data[isinf] = 0
    torch.nn.utils.clip_grad_norm_(model.parameters(), clip)
and PyTorch's ``clip_grad_norm_`` (third party; torch 2.10 in this image, same algorithm since 1.x, norm_type=2):
    total_norm = || (||g_1||_2, ..., ||g_n||_2) ||_2 ;  coef = min(1, max_norm / (total_norm + 1e-6)) ;  g_i *= coef.
Pinned in tests/test_gradclip.py against those very torch calls executed on CPU tensors.
"""
import numpy as np


def sanitize_and_clip(grads, max_norm):
    """grads: list of float32 numpy arrays (any shapes).  Returns (list of new float32 arrays, total_norm float64)."""
    clean = []
    sq = 0.0
    for g in grads:
        g = np.asarray(g, dtype=np.float32)
        c = np.where(np.isfinite(g), g, np.float32(0.0)).astype(np.float32)
        clean.append(c)
        sq += float(np.sum(c.astype(np.float64) ** 2))
    total = np.sqrt(sq)
    coef = np.float32(max_norm) / (np.float32(total) + np.float32(1e-6))
    coef = np.float32(min(coef, np.float32(1.0)))
    if coef < 1.0:
        clean = [(c * coef).astype(np.float32) for c in clean]
    return clean, total
