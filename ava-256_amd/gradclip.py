"""Multi-tensor gradient hygiene of the training loop on the HIP library.

Counterpart of /root/reference/ddp-train.py:434-441 -- per parameter ``grad[isnan] = 0; grad[isinf] = 0`` followed by
``torch.nn.utils.clip_grad_norm_(params, clip)`` -- as two kernel passes over all gradients and no host
synchronisation: the clip coefficient is computed on the device.  There is no CPU path.
"""
import ctypes

import torch

from . import _lib


class GradClipper:
    """Reusable: owns the two scalars the kernels need (a device double for the sum of squares, a device float for
    the norm).  ``clipper(params_or_grads, max_norm)`` returns the total norm as a 0-d device tensor."""

    def __init__(self, device):
        self.device = torch.device(device)
        self._sq = torch.zeros(1, dtype=torch.float64, device=self.device)
        self._norm = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._cache = None  # (addresses + sizes, pointer table, element counts) of the previous call

    @staticmethod
    def _grads_of(items):
        out = []
        for t in items:
            g = t.grad if isinstance(t, torch.nn.Parameter) or (hasattr(t, "grad") and t.requires_grad) else t
            if g is None:
                continue
            if g.dtype != torch.float32 or not g.is_cuda:
                raise RuntimeError("GradClipper: gradients must be float32 tensors on the GPU (got %s on %s)" %
                                   (g.dtype, g.device))
            if not g.is_contiguous():
                raise RuntimeError("GradClipper: gradients must be dense (contiguous)")
            out.append(g)
        return out

    @torch.no_grad()
    def __call__(self, params_or_grads, max_norm):
        grads = self._grads_of(params_or_grads)
        lib = _lib.get_lib()
        n = len(grads)
        # the ctypes tables are rebuilt only when an address or a size changed (zero_grad(set_to_none=False) and DDP's
        # bucket views keep them)
        key = [g.data_ptr() for g in grads] + [g.numel() for g in grads]
        c = self._cache
        if c is None or c[0] != key:
            ptrs = (ctypes.c_void_p * max(n, 1))(*key[:n])
            numels = (ctypes.c_longlong * max(n, 1))(*key[n:])
            self._cache = c = (key, ptrs, numels)
        ptrs, numels = c[1], c[2]
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            _lib.check(lib.mvp_grads_sanitize_sqnorm(n, ptrs, numels, self._sq.data_ptr(), stream),
                       "mvp_grads_sanitize_sqnorm")
            _lib.check(lib.mvp_grads_clip_scale(n, ptrs, numels, self._sq.data_ptr(), float(max_norm),
                                                self._norm.data_ptr(), stream), "mvp_grads_clip_scale")
        return self._norm[0].clone()  # a fresh scalar like clip_grad_norm_'s (the buffer is overwritten by the next call)


def sanitize_and_clip_(params_or_grads, max_norm):
    """One-shot form of GradClipper (allocates its two scalars per call)."""
    items = list(params_or_grads)
    dev = None
    for t in items:
        g = t.grad if hasattr(t, "grad") and t.grad is not None else t
        if torch.is_tensor(g) and g.is_cuda:
            dev = g.device
            break
    if dev is None:
        raise RuntimeError("sanitize_and_clip_: no GPU gradient given (there is no CPU path)")
    return GradClipper(dev)(items, max_norm)
