"""Fused background MLP (second half): autograd binding of mvp_bgmlp_forward / mvp_bgmlp_backward.

The per-pixel stack of /root/reference/models/bg/mlp2d.py:29-41 (1x1 convolutions 120 -> 256 x 5 -> 3, LeakyReLU(0.2))
runs as one MFMA kernel per direction that keeps a 128-pixel tile's activations in LDS across all layers
(csrc/bgmlp.hip).  The kernels produce the output and the chain of input gradients; weight and bias gradients are
[256 x P] . [P x 256] GEMMs and column sums over the stored bf16 (activation, gradient) pairs, issued here."""
import math

import torch

from . import _lib
from ._tensors import ptr, stream_ptr

WIDTH, POS, POS_PAD, HIDDEN, TILE = 256, 40, 48, 4, 256


def positional_encoding(samplecoords: torch.Tensor) -> torch.Tensor:
    """mlp2d.py:64-68: cat([sin(2^i pi x) for i < 10] + [cos(2^i pi x) for i < 10], -1) -> [..., 40]."""
    return torch.cat([torch.sin(2 ** i * math.pi * samplecoords) for i in range(10)] +
                     [torch.cos(2 ** i * math.pi * samplecoords) for i in range(10)], dim=-1)


def _chunks(M: int, chunks: int = 64) -> int:
    return chunks if M % chunks == 0 and M >= chunks * 256 else 1


def _wgrad(dy: torch.Tensor, x: torch.Tensor, chunks: int = 64) -> torch.Tensor:
    """dy^T @ x for tall bf16 matrices [M, a], [M, b] -- or stacks of them [L, M, a], [L, M, b] -> [L, a, b] -- with the
    M-reduction split into `chunks` batched GEMMs (a 256 x 256 output with K = M gives hipBLASLt 16 workgroups
    otherwise).  ONE bmm call for the whole stack, partial products delivered in fp32 (`out_dtype`) and summed in fp32:
    against four calls with bf16 partial products 0.76 instead of 1.20 ms at 4 x 512^2 pixels and 2e-7 instead of 1.7e-3
    from float64 (tools/bench_wgrad.py, profiles/r03_wgrad_bench.txt)."""
    stack = dy.dim() == 3
    L, M = (dy.shape[0], dy.shape[1]) if stack else (1, dy.shape[0])
    S = _chunks(M, chunks)
    r = torch.bmm(dy.reshape(L * S, M // S, -1).transpose(1, 2), x.reshape(L * S, M // S, -1), out_dtype=torch.float32)
    r = r.view(L, S, r.shape[-2], r.shape[-1]).sum(1)
    return r if stack else r[0]


class _FusedBgMlp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, train, samplecoords, bias1, w1pos, w6, b6, *hidden):
        # hidden = (W2, b2, W3, b3, W4, b4, W5, b5): weights [256,256] ([out][in]), biases [256]
        if not samplecoords.is_cuda:
            raise RuntimeError("fused background MLP: CUDA/HIP tensors only (no CPU fallback)")
        if samplecoords.dim() != 4 or samplecoords.shape[-1] != 2:
            raise ValueError("samplecoords must be [B,H,W,2]")
        B, H, W = samplecoords.shape[:3]
        if (len(hidden) != 2 * HIDDEN or tuple(w1pos.shape) != (WIDTH, POS) or tuple(w6.shape) != (3, WIDTH)
                or tuple(bias1.shape) != (B, WIDTH) or tuple(b6.shape) != (3,)
                or any(tuple(hidden[2 * i].shape) != (WIDTH, WIDTH) or tuple(hidden[2 * i + 1].shape) != (WIDTH,)
                       for i in range(HIDDEN))):
            raise ValueError("fused background MLP: the kernels are built for 40 -> 256 x 5 -> 3 (mlp2d.py:29-41)")
        HW, dev = H * W, samplecoords.device
        sc = samplecoords.detach().float().contiguous()
        w1p = torch.zeros((WIDTH, POS_PAD), device=dev, dtype=torch.bfloat16)
        w1p[:, :POS] = w1pos.detach()
        whb = torch.stack([hidden[2 * i].detach() for i in range(HIDDEN)]).to(torch.bfloat16).contiguous()
        bhf = torch.stack([hidden[2 * i + 1].detach() for i in range(HIDDEN)]).float().contiguous()
        b1f, w6f, b6f = bias1.detach().float().contiguous(), w6.detach().float().contiguous(), b6.detach().float().contiguous()
        # `train` (fused_background_mlp): grad mode is on and something requires a gradient.  ctx.needs_input_grad alone
        # also says True under torch.no_grad() -- the inference kernel (no stores at all) would never run
        need_grad = bool(train) and any(ctx.needs_input_grad)
        acts = torch.empty((HIDDEN + 1, B * HW, WIDTH), device=dev, dtype=torch.bfloat16) if need_grad else None
        x0 = torch.empty((B * HW, POS_PAD), device=dev, dtype=torch.bfloat16) if need_grad else None
        out = torch.empty((B, 3, H, W), device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(_lib.get_lib().mvp_bgmlp_forward(B, HW, ptr(sc), ptr(b1f), ptr(w1p), ptr(whb), ptr(bhf), ptr(w6f),
                                                        ptr(b6f), ptr(acts), ptr(x0), ptr(out), stream_ptr(dev)),
                       "mvp_bgmlp_forward")
        ctx.save_for_backward(x0, whb, w6f, acts)
        ctx.dims = (B, H, W)
        return out

    @staticmethod
    def backward(ctx, gout):
        x0, whb, w6f, acts = ctx.saved_tensors
        B, H, W = ctx.dims
        HW, dev = H * W, x0.device
        P, tiles = B * HW, (HW + TILE - 1) // TILE
        gout = gout.float().contiguous()
        whT = whb.transpose(1, 2).contiguous()
        dz = torch.empty((HIDDEN + 1, P, WIDTH), device=dev, dtype=torch.bfloat16)
        colsum = torch.empty((HIDDEN + 1, B, tiles, WIDTH), device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(_lib.get_lib().mvp_bgmlp_backward(B, HW, ptr(gout), ptr(acts), ptr(whT), ptr(w6f), ptr(dz),
                                                         ptr(colsum), stream_ptr(dev)), "mvp_bgmlp_backward")
        g_bias1 = colsum[0].sum(1)
        g_w1pos = _wgrad(dz[0], x0)[:, :POS]
        # last layer (mlp2d.py:69: output = MLP * 25 + 100).  gout is [B,3,HW]: its rows ARE the [3 x pixels] operand of the
        # weight gradient, image by image, and the bias gradient is a sum over contiguous planes (the transposed [P,3]
        # copy and its strided column sum cost 0.36 ms at 4 x 512^2)
        S6 = _chunks(HW, 16)
        g6 = (gout.view(B, 3, S6, HW // S6) * 25.0).to(torch.bfloat16).permute(0, 2, 1, 3).reshape(B * S6, 3, HW // S6)
        g_w6 = torch.bmm(g6, acts[HIDDEN].view(B * S6, HW // S6, WIDTH), out_dtype=torch.float32).sum(0)
        g_b6 = gout.sum((0, 2, 3)) * 25.0
        g_wh = _wgrad(dz[1:], acts[:HIDDEN])
        g_bh = colsum[1:].sum((1, 2))
        hidden = []
        for l in range(HIDDEN):
            hidden += [g_wh[l], g_bh[l]]
        return (None, None, g_bias1, g_w1pos, g_w6, g_b6, *hidden)


IMAGES_PER_CALL = 16  # training: images per kernel pair (see fused_background_mlp)


def fused_background_mlp(samplecoords, bias1, w1pos, hidden, w6, b6, images_per_call=None):
    """samplecoords [B,H,W,2]; bias1 [B,256] (first-layer bias incl. the camera / identity codes); w1pos [256,40];
    hidden = [(W, b)] x 4; w6 [3,256]; b6 [3]  ->  [B,3,H,W] = MLP * 25 + 100 (mlp2d.py:69-70).

    Training keeps 2.6 KB per pixel for the backward (five bf16 activation planes + the positional encoding) and the
    backward makes as much again (the five gradient planes the weight-gradient GEMMs read): at 80 x 512^2 pixels 2 x 54 GB
    if everything is one call.  A batch is therefore walked in groups of `images_per_call` images (default
    IMAGES_PER_CALL), each its own autograd node: the activation planes of all groups stay until their backward, the
    gradient planes exist for one group at a time (80 frames: 54 + 11 GB).  Tiles never span images, so the output bits do
    not depend on the grouping; weight gradients are summed over the groups in fp32 by autograd."""
    flat = [t for wb in hidden for t in wb]
    train = torch.is_grad_enabled() and any(t.requires_grad for t in (bias1, w1pos, w6, b6, *flat))
    # Pixel coordinates are data (mlp2d.py:56-60 receives them from the batch): no gradient is defined for them here.  Detached,
    # so that a samplecoords tensor that happens to require grad cannot make autograd build a node over an inference-mode
    # forward (which saves nothing for a backward).
    samplecoords = samplecoords.detach()
    B = samplecoords.shape[0]
    n = int(images_per_call or IMAGES_PER_CALL)
    if not train or B <= n:
        return _FusedBgMlp.apply(train, samplecoords, bias1, w1pos, w6, b6, *flat)
    return torch.cat([_FusedBgMlp.apply(train, samplecoords[i:i + n], bias1[i:i + n], w1pos, w6, b6, *flat)
                      for i in range(0, B, n)], dim=0)
